#!/usr/bin/env python3
"""Summarise rocprofv3 --kernel-trace CSVs: per (kernel, grid size) the number of dispatches and their mean / min duration in us."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(list)
for pat in sys.argv[1:]:
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'][:60], int(r['Grid_Size_X']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (k, g), v in sorted(agg.items()):
    if 'gemm' not in k and 'attention' not in k and 'attn' not in k:
        continue
    v = sorted(v)
    n = len(v)
    core = v[: max(1, n * 3 // 4)]            # drop the slowest quarter (first launches, clock ramps)
    print(f'{k:60s} grid={g:8d} wgs={g // 256:5d} n={n:4d} mean={sum(v) / n:8.1f} us  fastest-3/4 mean={sum(core) / len(core):8.1f} us  min={v[0]:8.1f}')
