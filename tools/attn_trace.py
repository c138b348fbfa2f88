#!/usr/bin/env python3
"""s_memtime stamps of one KV iteration of the 8-wave attention kernel (needs a library built with -DAFX_ATTN_TRACE:
   AFX_EXTRA_FLAGS=-DAFX_ATTN_TRACE python -m arcflow_amd.build --force)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import _lib, ops  # noqa: E402

lib = _lib.load()
q, k, v = (torch.randn(1, 4608, 24, 128, device='cuda').bfloat16() for _ in range(3))
for _ in range(3):
    ops.attention(q, k, v)
torch.cuda.synchronize()
buf = (C.c_uint * 128)()
assert lib.afx_debug_attn_trace(buf) == 0, 'library was not built with -DAFX_ATTN_TRACE'
names = ['M:Kread+PV', 'QK', 'lgkm+bar', '', 'Vread+DMA', 'softmax', 'vmcnt', 'bar']
for blk in range(2):
    print(f'block {"0" if blk == 0 else "1000"}: per wave  [M: K reads + PV | QK | wait + barrier] [S: V reads + DMA issue | softmax | vmcnt wait | barrier]')
    for w in range(8):
        t = [buf[(blk * 8 + w) * 8 + i] for i in range(8)]
        d = [(t[i + 1] - t[i]) & 0xffffffff for i in range(7)]
        print(f'  w{w}: M {d[0]:5d} {d[1]:5d} {d[2]:5d} | S {d[3]:5d} {d[4]:5d} {d[5]:5d} {d[6]:5d} | total {sum(d):6d}   (start {t[0] - buf[(blk*8)*8]:+d})')

# ---- 4-wave kernel (the default): whole-work-group cycles and the shader clock it ran at, after a settled run of launches
if hasattr(lib, 'afx_debug_attn_trace4'):
    for _ in range(200):
        ops.attention(q, k, v)
    torch.cuda.synchronize()
    b4 = (C.c_uint * 32)()
    if lib.afx_debug_attn_trace4(b4) == 0:
        for blk in range(2):
            for w in range(4):
                cyc, ticks, tiles, _ = (b4[(blk * 4 + w) * 4 + i] for i in range(4))
                print(f'4-wave kernel block {"0" if blk == 0 else "700"} w{w}: {cyc} cycles for {tiles} KV tiles = {cyc / max(tiles, 1):.0f} / tile'
                      f' (MFMA floor 1024), {ticks} ticks of 10 ns -> shader clock {100.0 * cyc / max(ticks, 1):.0f} MHz')

# ---- one-wave-per-SIMD kernel (afx_attn3.hip): iteration 36 of two work-groups
if hasattr(lib, 'afx_debug_attn3_trace'):
    ops.set_attn_impl(0)
    for _ in range(200):
        ops.attention(q, k, v)
    torch.cuda.synchronize()
    b3 = (C.c_uint * 128)()
    if lib.afx_debug_attn3_trace(b3) == 0:
        for blk in range(2):
            for w in range(4):
                x = [b3[(blk * 4 + w) * 16 + i] for i in range(8)]
                cyc, ticks, tiles = x[0], x[1], x[2]
                print(f'v3 kernel block {"0" if blk == 0 else "300"} w{w}: {cyc} cycles for {tiles} KV tiles = {cyc / max(tiles, 1):.0f} / tile (MFMA floor 2048),'
                      f' clock {100.0 * cyc / max(ticks, 1):.0f} MHz | iteration 36: wait {x[3]} barrier {x[4]} phase A {x[5]} phase B {x[6]}')
