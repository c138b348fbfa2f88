#!/usr/bin/env python3
"""Flash-attention backward at the production shape (S = 4608 = 4096 image + 512 text tokens, 24 heads x 128, one sample): time per
call (all of afx_attention_backward: 3 transposes + delta + dQ kernel + dK/dV kernel) and TFLOP/s over the 5 algorithmic matmuls."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import ops  # noqa: E402

S, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 4608), 24
g = torch.Generator(device='cuda').manual_seed(0)
q, k, v, do = (torch.randn(1, S, H, 128, generator=g, device='cuda').bfloat16() for _ in range(4))
o, lse = ops.attention_fwd_lse(q, k, v)
for _ in range(3):
    ops.attention_bwd(q, k, v, o, do, lse)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    ops.attention_bwd(q, k, v, o, do, lse)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / n * 1e6
fl = 5 * 2 * H * S * S * 128
print(f'attention backward S={S} H={H}: {us:.1f} us per call = {fl / us * 1e-6:.0f} TFLOP/s over 5 matmuls ({fl / us * 1e-6 / 2500:.3f} of 2.5 PF)')
