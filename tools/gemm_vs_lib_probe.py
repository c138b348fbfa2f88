#!/usr/bin/env python3
"""Same operands through gemm_kernel_v3 and through torch (hipBLASLt), N launches each -- for rocprofv3 kernel-trace / --pmc passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import ops  # noqa: E402

shapes = [(8192, 8192, 8192), (4608, 21504, 3072), (4608, 3072, 15360)]
n = int(os.environ.get('N', '12'))
if os.environ.get('SHAPE'):
    shapes = [shapes[int(os.environ['SHAPE'])]]
for M, N, K in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16()
    w = (torch.randn(N, K, device='cuda') * 0.02).bfloat16()
    b = torch.randn(N, device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(n):
        ops.linear(a, w, b, out=out)
    torch.cuda.synchronize()
    for _ in range(n):
        torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
