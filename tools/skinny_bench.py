#!/usr/bin/env python3
"""Rank-256 LoRA products of the distillation step (N = 256, long K): plain launch vs split-K slabs + fold, us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K in [(4096, 256, 3072), (512, 256, 3072), (4608, 256, 3072), (4096, 256, 12288), (512, 256, 12288), (4608, 256, 15360), (4608, 256, 12288)]:
    a = torch.randn(M, K, device='cuda').bfloat16()
    w = (torch.randn(N, K, device='cuda') * 0.02).bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    t0 = timeit(lambda: ops.linear(a, w, out=out))
    res = [f'plain {t0:6.1f}']
    for sk in (0, 4, 8, 16):
        res.append(f'sk{sk} {timeit(lambda: ops.linear_splitk(a, w, out=out, split_k=sk)):6.1f}')
    ref = a.float() @ w.float().t()
    err = ((ops.linear_splitk(a, w).float() - ref).norm() / ref.norm()).item()
    print(f'M={M:5d} N={N} K={K:5d}: ' + '  '.join(res) + f'   rel err {err:.1e}')
