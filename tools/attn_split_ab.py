#!/usr/bin/env python3
"""Attention launch time with the balanced last round (impl 0: short ends first, long parts continue from their partials) against the plain grid
(impl 3), interleaved, at the FLUX / Qwen-Image joint shapes.  Times transpose + attention (+ combine) through the C ABI; kernel-level split: rocprofv3 --kernel-trace of this script."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from arcflow_amd import ops  # noqa: E402


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


g = torch.Generator(device='cuda').manual_seed(1)
shapes = ((1, 4608, 24), (1, 4224, 24), (1, 4173, 24), (2, 4608, 24), (4, 4608, 24))
if len(sys.argv) > 1:
    shapes = (tuple(int(x) for x in sys.argv[1].split(',')),)
for B, S, H in shapes:
    q, k, v = (torch.randn(B, S, H, 128, generator=g, device='cuda').bfloat16() for _ in range(3))
    q2, k2, v2 = (t.reshape(B * S, H * 128) for t in (q, k, v))
    o = torch.empty_like(q2)
    for rnd in range(2):
        for impl, name in ((3, 'plain grid'), (0, 'balanced  ')):
            ops.set_attn_impl(impl)
            us = timed(lambda: ops.attention_fwd_lse_2d(q2, k2, v2, o, B, S, H))
            print(f'B={B} S={S} H={H} {name}: {us:7.1f} us  (V transpose + attention + lse fill)  {4 * B * H * S * S * 128 / us / 1e6:6.0f} TF/s', flush=True)
ops.set_attn_impl(0)
