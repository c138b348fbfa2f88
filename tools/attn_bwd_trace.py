#!/usr/bin/env python3
"""Cycle stamps of the dK / dV kernel's five scheduling regions (query tile 36 of two work-groups) and whole-kernel cycles per tile.
Needs the trace variant:  python -m arcflow_amd.build --variant bwdtrace -DAFX_BWD_TRACE -- afx_attn_bwd.hip   and
ARCFLOW_HIP_LIB=arcflow_amd/lib/libarcflow_hip_bwdtrace.so python tools/attn_bwd_trace.py"""
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
for _ in range(50):
    ops.attention_bwd(q, k, v, o, do, lse)
torch.cuda.synchronize()
buf = (C.c_uint * 96)()
lib.afx_debug_bwd_trace.argtypes = [C.c_void_p]
assert lib.afx_debug_bwd_trace(buf) == 0, 'library was not built with -DAFX_BWD_TRACE'
for blk in range(2):
    for w in range(4):
        t = [buf[(blk * 4 + w) * 12 + i] for i in range(12)]
        d = [(t[i + 1] - t[i]) & 0xffffffff for i in range(6)]
        cyc, ticks, tiles = t[8], t[9], t[11]
        print(f'block {"0" if blk == 0 else "1000"} w{w}: R0 {d[0]:5d} R1 {d[1]:5d} R2 {d[2]:5d} R3 {d[3]:5d} R4 {d[4]:5d} wait+barrier {d[5]:5d} | sum {sum(d):6d} | '
              f'kernel {cyc} cycles / {tiles} tiles = {cyc / max(tiles, 1):.0f} per tile (MFMA floor 2048), clock {100.0 * cyc / max(ticks, 1):.0f} MHz')
