#!/usr/bin/env python3
"""A/B of the one-wave-per-SIMD GEMM's tile shapes on the forward's shapes (isolated launches, weights warm in the Infinity Cache):
256x256 (tile mode 1) against 256x224 (mode 5) and the launcher's own choice (mode 0), interleaved rounds, median."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import ops  # noqa: E402

SHAPES = [(4608, 3072, 3072, 'out / single-round'), (4608, 3072, 12288, 'mlp2'), (4608, 3072, 15360, 'single out'), (4608, 12288, 3072, 'mlp1'),
          (4608, 9216, 3072, 'qkv (no fusion here)'), (4608, 21504, 3072, 'single fused')]
g = torch.Generator(device='cuda').manual_seed(0)
for M, N, K, what in SHAPES:
    a = torch.randn(M, K, generator=g, device='cuda').bfloat16()
    w = (torch.randn(N, K, generator=g, device='cuda') * 0.02).bfloat16()
    b = torch.randn(N, generator=g, device='cuda').bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    res = {}
    for rnd in range(5):
        for mode in (1, 5, 0):
            ops.set_gemm_mode(3, mode)
            for _ in range(3):
                ops.linear(a, w, b, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.linear(a, w, b, out=out)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) * 1e3 / 20)
    med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
    tf = {m: 2.0 * M * N * K / us * 1e-6 for m, us in med.items()}
    print(f'{M}x{N}x{K} {what:22s} 256x256 {med[1]:7.1f} us {tf[1]:6.0f} TF | 256x224 {med[5]:7.1f} us {tf[5]:6.0f} TF ({100 * (med[1] / med[5] - 1):+.1f} %) | auto {med[0]:7.1f} us', flush=True)
ops.set_gemm_mode(3, 0)
