#!/usr/bin/env python3
"""Repeat each hot kernel on fixed inputs and count launches whose output is not bit-identical to the first one.
A correct kernel gives 0 everywhere; a rare LDS / DMA ordering bug shows up as a few differing launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import ops  # noqa: E402

REPS = int(os.environ.get('REPS', 300))


def count(fn, reps=REPS, disturb=None):
    ref = fn().clone()
    bad = 0
    worst = 0.0
    for i in range(reps):
        if disturb is not None and i % 3 == 0:
            disturb()
        out = fn()
        if not torch.equal(out, ref):
            bad += 1
            worst = max(worst, (out.float() - ref.float()).abs().max().item())
    return bad, worst


def main():
    g = torch.Generator(device='cuda').manual_seed(0)
    junk = torch.empty(64 << 20, dtype=torch.float32, device='cuda')
    disturb = lambda: junk.normal_()  # noqa: E731  (256 MB of writes: evicts L2 / MALL, changes the timing)
    for M, N, K in [(4608, 3072, 3072), (4608, 9216, 3072), (4608, 21504, 3072), (4608, 3072, 15360), (512, 9216, 3072)]:
        a = torch.randn(M, K, generator=g, device='cuda').bfloat16()
        w = (torch.randn(N, K, generator=g, device='cuda') * 0.02).bfloat16()
        b = torch.randn(N, generator=g, device='cuda').bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
        print(f'gemm {M}x{N}x{K} plain   :', count(lambda: ops.linear(a, w, b, out=out), disturb=disturb), flush=True)
        ws = ops.stream_k_workspace()
        print(f'gemm {M}x{N}x{K} stream-K:', count(lambda: ops.linear(a, w, b, out=out, sk_ws=ws), disturb=disturb), flush=True)
    for S in (4608, 4173):
        q, k, v = (torch.randn(1, S, 24, 128, generator=g, device='cuda').bfloat16() for _ in range(3))
        print(f'attention S={S}:', count(lambda: ops.attention(q, k, v), reps=REPS // 2, disturb=disturb), flush=True)
    x = torch.randn(4608, 3072, generator=g, device='cuda').bfloat16()
    sc, sh = torch.randn(1, 3072, generator=g, device='cuda'), torch.randn(1, 3072, generator=g, device='cuda')
    print('norm_modulate:', count(lambda: ops.norm_modulate(x, sc, sh), disturb=disturb), flush=True)
    xw = torch.randn(1, 3072, generator=g, device='cuda')
    W = torch.randn(132096, 3072, generator=g, device='cuda').bfloat16()
    print('gemv:', count(lambda: ops.gemv(xw, W, None), reps=REPS // 3, disturb=disturb), flush=True)


if __name__ == '__main__':
    print({k: v for k, v in os.environ.items() if k.startswith('AFX_')})
    main()
