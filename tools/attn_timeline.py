#!/usr/bin/env python3
"""When did each attention work-group run?  (afx_debug_attn_timeline: start / end of every work-group in the 100 MHz realtime counter.)  Prints, for
XCD 0 of the balanced schedule and of the plain grid, the work-groups in dispatch order with start / end in us and the segments they ran."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from arcflow_amd import ops, _lib  # noqa: E402

B, S, H = (int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else (1, 4608, 24)))
lib = _lib.load()
g = torch.Generator(device='cuda').manual_seed(1)
q, k, v = (torch.randn(B, S, H, 128, generator=g, device='cuda').bfloat16() for _ in range(3))
W = 1 + 8 * 8
items = (C.c_int * (W * 16384))()
nparts, grid = C.c_int(0), C.c_int(0)
lib.afx_debug_attn_plan.restype = C.c_int
n = lib.afx_debug_attn_plan(B, H, S, 256, items, 16384, C.byref(nparts), C.byref(grid))
plan = np.frombuffer(items, dtype=np.int32)[:W * grid.value].reshape(grid.value, W) if n > 0 else None
for impl, name in ((3, 'plain grid'), (0, 'balanced')):
    ops.set_attn_impl(impl)
    for _ in range(3):
        ops.attention(q, k, v)
    torch.cuda.synchronize()
    assert lib.afx_debug_attn_timeline(1) == 0
    ops.attention(q, k, v)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (4 * 8192))()
    lib.afx_debug_attn_timeline_read.restype = C.c_int
    ng = lib.afx_debug_attn_timeline_read(buf, 8192)
    lib.afx_debug_attn_timeline(0)
    t = np.frombuffer(buf, dtype=np.uint64)[:4 * ng].reshape(ng, 4).astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    print(f'== {name}: {ng} work-groups, makespan {us[:, 3].max():.1f} us; per XCD: ' + ' '.join(f'{us[x::8, 3].max():.0f}' for x in range(8)))
    for i in range(0, ng, 8):
        desc = ''
        if impl == 0 and plan is not None:
            ns = plan[i, 0]
            desc = ' | '.join(f'h{plan[i, 1 + 8 * s_]} q{plan[i, 3 + 8 * s_]} [{plan[i, 4 + 8 * s_]},+{plan[i, 5 + 8 * s_]}) out {plan[i, 6 + 8 * s_]} in {plan[i, 7 + 8 * s_]}' for s_ in range(ns))
        print(f'{i // 8:3d} se{(i // 8) % 4}  {us[i, 0]:7.1f} {us[i, 3]:7.1f}  ({us[i, 3] - us[i, 0]:6.1f}; last segment: prologue issued +{us[i, 1] - us[i, 0]:5.1f}, loop entered +{us[i, 2] - us[i, 0]:5.1f})  {desc}')
ops.set_attn_impl(0)
