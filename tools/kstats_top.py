#!/usr/bin/env python3
"""Top rows of a rocprofv3 --stats kernel_stats.csv:  python tools/kstats_top.py <csv> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:n]:
    print(r['Name'][:72].ljust(72), r['Calls'].rjust(6), f"{float(r['AverageNs']) / 1e3:9.1f} us", f"{float(r['TotalDurationNs']) / 1e6:8.1f} ms",
          f"{100 * float(r['TotalDurationNs']) / tot:5.1f} %")
