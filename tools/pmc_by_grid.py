#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per (kernel, grid size), mean counter value per dispatch."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for pat in sys.argv[1:]:
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'][:70], r.get('Grid_Size', '?'))][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), v in sorted(agg.items()):
    if 'gemm' not in k and 'Cijk' not in k and 'attention' not in k and 'attn_bwd' not in k:
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    util = ''
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and m.get('GRBM_GUI_ACTIVE', 0) > 0:      # both are sums over their instances: 8 XCDs, 1024 SIMDs
        util = f"  -> MFMA utilisation {100.0 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.1f} %"
    print(f'{k} grid={g} n={max(len(x) for x in v.values())}  ' + '  '.join(f'{c}={x:.0f}' for c, x in sorted(m.items())) + util)
