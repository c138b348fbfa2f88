"""Group a rocprofv3 --kernel-trace CSV by (kernel, grid size): launches, mean duration.  The GEMM grid size identifies the shape
(216 / 256 / 648 / 864 / 1512 ... tiles x threads), so two runs with different GEMM kernels can be compared shape by shape."""
import csv
import glob
import sys
from collections import defaultdict

rows = defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('void ', '')
        name = name.split('(')[0] if name.startswith('afx::') else name[:150]
        wg = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1)
        grid = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)
        rows[(name, grid // max(wg, 1))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in rows.values())
for (name, wgs), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / tot < 0.004:
        continue
    print(f'{name[:150]:44s} wgs={wgs:6d} n={len(v):5d} mean={sum(v)/len(v):9.1f} us  share={100*sum(v)/tot:5.1f} %')
