#!/usr/bin/env python3
"""Isolated timing of the library's bf16 linear on the FLUX GEMM shapes (+ 8192^3): TFLOP/s per shape, one line.  For A/B runs of
variant libraries (ARCFLOW_HIP_LIB=...)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import ops  # noqa: E402

shapes = [(4608, 9216, 3072), (4608, 3072, 3072), (4608, 12288, 3072), (4608, 3072, 12288), (4608, 21504, 3072), (4608, 3072, 15360),
          (8192, 8192, 8192)]
out_s = []
for M, N, K in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16()
    w = (torch.randn(N, K, device='cuda') * 0.02).bfloat16()
    b = torch.randn(N, device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    fn = (lambda: torch.nn.functional.linear(a, w, b)) if os.environ.get('TORCH') else (lambda: ops.linear(a, w, b, out=out))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / 20 * 1e-3
    out_s.append(f'{2 * M * N * K / dt / 1e12:6.0f}')
print(' '.join(out_s))
