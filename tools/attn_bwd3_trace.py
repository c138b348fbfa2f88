#!/usr/bin/env python3
"""Cycle stamps of eight consecutive phases (p = 72..79) of the generated dK / dV kernel (afx_attn_bwd3.hip), two work-groups, per wave:
phase start -> behind the counted wait + barrier -> behind MFMA gap 15 (the S / dP half) -> next phase start, and whole-kernel cycles per phase.
Needs the trace variant:  python -m arcflow_amd.build --variant bwd3trace -DAFX_BWD3_TRACE -- afx_attn_bwd3.hip   and
ARCFLOW_HIP_LIB=arcflow_amd/lib/libarcflow_hip_bwd3trace.so python tools/attn_bwd3_trace.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import _lib, ops  # noqa: E402

lib = _lib.load()
S, H = 4608, 24
g = torch.Generator(device='cuda').manual_seed(0)
q, k, v, do = (torch.randn(1, S, H, 128, generator=g, device='cuda').bfloat16() for _ in range(4))
o, lse = ops.attention_fwd_lse(q, k, v)
for _ in range(20):
    ops.attention_bwd(q, k, v, o, do, lse)
torch.cuda.synchronize()
buf = (C.c_uint * 512)()
lib.afx_debug_bwd3_trace.argtypes = [C.c_void_p]
assert lib.afx_debug_bwd3_trace(buf) == 0, 'library was not built with -DAFX_BWD3_TRACE'
for kern, blk in ((0, 0), (0, 1), (1, 0), (1, 1)):
    floor = (1024, 768)[kern]
    for w in range(4):
        t = [buf[((kern * 2 + blk) * 4 + w) * 32 + i] for i in range(32)]
        cyc, nh, ticks = t[24], t[25], t[26]
        ph = []
        for j in range(7):
            a, b, c, n = t[3 * j], t[3 * j + 1], t[3 * j + 2], t[3 * j + 3]
            ph.append(f'{(b - a) & 0xffffffff:4d}+{(c - b) & 0xffffffff:4d}+{(n - c) & 0xffffffff:4d}')
        print(f'{("dkv3", "dq3")[kern]} block {"0" if blk == 0 else "800"} w{w}: wait/barrier + S/dP half + accumulate half per phase: ' + ' | '.join(ph) +
              f' || kernel {cyc} cycles / {nh + 2} phases = {cyc / (nh + 2):.0f} per phase (MFMA floor {floor}), shader clock {100.0 * cyc / max(ticks, 1):.0f} MHz')
