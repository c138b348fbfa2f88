#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, mean counter value per dispatch."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    n = max(len(x) for x in v.values())
    print(f'{k}  dispatches={n}')
    for c, vals in sorted(v.items()):
        print(f'   {c:32s} {sum(vals)/len(vals):18.0f}')
