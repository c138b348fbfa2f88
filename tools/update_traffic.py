#!/usr/bin/env python3
"""profiles/traffic.json from the two PMC passes of tools/profile_round.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs of
`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile`): dispatch-weighted mean over the block GEMM kernel's instantiations,
corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes -- FETCH_SIZE is in KiB and reads exactly 1/2 of a wide (16 B / lane)
coalesced stream on gfx950: bytes = (2 FETCH_SIZE + WRITE_SIZE) x 1024.  The file is stamped with the sha of the kernel sources so that bench.py
prints `traffic` only for the build the pass measured.
usage: python tools/update_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <round tag> [model]"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and 'gemm_kernel_v3<' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0].replace('void afx::', '').strip()].append(float(r['Counter_Value']))
    return agg


def main():
    fetch, write, tag = sys.argv[1:4]
    model = sys.argv[4] if len(sys.argv) > 4 else 'flux'
    import bench
    f, w = per_kernel(fetch, 'FETCH_SIZE'), per_kernel(write, 'WRITE_SIZE')
    keep = {k for k in f if len(f[k]) >= 50}                      # the forward's block GEMMs (the 128x128 instance has a handful of launches)
    nf = sum(len(f[k]) for k in keep)
    fk = sum(sum(f[k]) for k in keep) / nf
    wk = sum(sum(w[k]) for k in keep if k in w) / sum(len(w[k]) for k in keep if k in w)
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        t = json.load(open(path))
    except Exception:
        t = {}
    t['_comment'] = ('HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of '
                     '`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile`), corrected as /opt/skills/guides/MI355X_MICROARCH.md HBM section '
                     'prescribes: FETCH_SIZE is in KiB and reads exactly 1/2 of a wide (16 B/lane) coalesced stream on gfx950 -> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024. '
                     'Infinity-Cache hits are counted (these are L2 memory-side requests). Written by tools/update_traffic.py; kernel_source_sha16 = the build it measured '
                     '(bench.py prints null for any other).')
    t['kernel_source_sha16'] = bench._kernel_source_sha()
    t[model] = {
        'kernel': ' + '.join(f'afx::{k} ({len(f[k])} dispatches)' for k in sorted(keep)) + ': every block GEMM launch of the forward',
        'round': tag, 'fetch_size_kib': round(fk), 'write_size_kib': round(wk), 'bytes_per_launch': int((2 * fk + wk) * 1024),
        'per_kernel': {k: {'fetch_size_kib': round(sum(f[k]) / len(f[k])), 'write_size_kib': round(sum(w[k]) / len(w[k])) if k in w else None} for k in sorted(keep)},
    }
    json.dump(t, open(path, 'w'), indent=1)
    print(json.dumps(t[model], indent=1))


if __name__ == '__main__':
    main()
