#!/usr/bin/env python3
"""The four GEMM shapes of a T5-XXL layer at 512 rows (two 256-row tiles): the split-K path text_encoders.py takes against the plain fused GEMM on
every tile shape of the one-wave-per-SIMD kernel.  us per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from arcflow_amd import ops  # noqa: E402


def timed(fn, reps=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


g = torch.Generator(device='cuda').manual_seed(0)
names = {0: 'auto', 1: '256x256', 2: '288x192', 3: '320x192', 4: '128x128', 5: '256x224', 6: '224x256'}
SHAPES = [('t5 qkv', 512, 12288, 4096), ('t5 o', 512, 4096, 4096), ('t5 wi', 512, 20480, 4096), ('t5 wo', 512, 4096, 10240),
          ('qwen qkv', 162, 4608, 3584), ('qwen o', 162, 3584, 3584), ('qwen gate|up', 162, 37888, 3584), ('qwen down', 162, 3584, 18944),
          ('clip qkv', 77, 2304, 768), ('clip fc1', 77, 3072, 768), ('clip fc2', 77, 768, 3072)]
for nm, M, N, K in SHAPES:
    a = torch.randn(M, K, generator=g, device='cuda').bfloat16()
    w = (torch.randn(N, K, generator=g, device='cuda') * 0.02).bfloat16()
    res = torch.randn(M, N, generator=g, device='cuda').bfloat16()
    line = [f'{nm:12s} M={M:3d} N={N:5d} K={K:5d}: split-K {timed(lambda: ops.linear_splitk(a, w, None, res)):6.1f} us']
    for sk in (2, 4, 8, 16):
        line.append(f'split-K x{sk} {timed(lambda: ops.linear_splitk(a, w, None, res, split_k=sk)):6.1f}')
    for tile in range(0, 7):
        ops.set_gemm_mode(3, tile)
        try:
            line.append(f'{names[tile]} {timed(lambda: ops.linear(a, w, None, epilogue="gate_res", residual=res)):6.1f}')
        except Exception as e:      # noqa: BLE001
            line.append(f'{names[tile]} n/a')
    ops.set_gemm_mode(3, 0)
    print('  '.join(line), flush=True)
