#!/usr/bin/env python3
"""The analytic ArcFlow step alone at 1024^2 (4096 tokens, K = 16, bf16 mixture): run under
`rocprofv3 --kernel-trace --stats` for the kernel's own duration (HIP-event timing includes ~4 us of launch gap)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import ops  # noqa: E402

B, N, K, ch, pp = 1, 4096, 16, 64, 4
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(B, N, ch, device='cuda', generator=g)
m = torch.randn(B, N, K, ch, device='cuda', generator=g).bfloat16()
lw = torch.log_softmax(torch.randn(B, N, K, pp, device='cuda', generator=g), 2).bfloat16()
lg = torch.randn(B, N, K - 1, pp, device='cuda', generator=g).bfloat16()
out = torch.empty_like(x)
junk = torch.empty(128 << 20, dtype=torch.float32, device='cuda')
for i in range(60):
    if i % 2 == 0:
        junk.zero_()                 # 512 MB of writes: the mixture is NOT left in L2 / MALL by the previous launch
    ops.arcflow_step(x, m, lw, lg, 1.0, 1.0, 0.7619, out=out)
torch.cuda.synchronize()
byts = x.numel() * 8 + (m.numel() + lw.numel() + lg.numel()) * 2
print(f'algorithmic bytes per launch: {byts} ({byts/1e6:.2f} MB)')
