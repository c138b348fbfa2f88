#!/usr/bin/env python3
"""normout_backward (d_scale / d_shift of an AdaLN site) at the training shape: us per call (AFX_NORMOUT_RPW sweeps the rows per wave)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import ops  # noqa: E402

for rows in (4096, 512, 4608):
    x = torch.randn(rows, 3072, device='cuda').bfloat16()
    d = torch.randn(rows, 3072, device='cuda').bfloat16()
    acc = torch.zeros(2 * 3072, device='cuda')
    for _ in range(3):
        ops.normout_backward(x, d, acc, rows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.normout_backward(x, d, acc, rows)
    e1.record()
    torch.cuda.synchronize()
    print(f'rows={rows}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us', end='   ')
print()
