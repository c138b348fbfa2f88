#!/usr/bin/env python3
"""Cycle stamps of the one-wave-per-SIMD attention kernel (ARCFLOW_HIP_LIB=arcflow_amd/lib/libarcflow_hip_trace.so, built by
`python -m arcflow_amd.build --variant trace -DAFX_ATTN_TRACE -- afx_attn3.hip afx_attn.hip`)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import _lib, ops  # noqa: E402

lib = _lib.load()
S, H = int(os.environ.get('S', '4608')), int(os.environ.get('H', '24'))
q, k, v = (torch.randn(1, S, H, 128, device='cuda').bfloat16() for _ in range(3))
ops.set_attn_impl(0)
for _ in range(200):
    ops.attention(q, k, v)
torch.cuda.synchronize()
b3 = (C.c_uint * 128)()
assert lib.afx_debug_attn3_trace(b3) == 0, 'library was not built with -DAFX_ATTN_TRACE'
for blk in range(2):
    for w in range(4):
        x = [b3[(blk * 4 + w) * 16 + i] for i in range(8)]
        cyc, ticks, tiles = x[0], x[1], x[2]
        print(f'block {"0" if blk == 0 else "300"} w{w}: {cyc} cycles for {tiles} KV tiles = {cyc / max(tiles, 1):.0f} / tile (MFMA floor 2048), '
              f'clock {100.0 * cyc / max(ticks, 1):.0f} MHz | iteration 36: wait {x[3]} barrier {x[4]} phase A {x[5]} phase B {x[6]}')
