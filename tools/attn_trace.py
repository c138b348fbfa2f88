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
