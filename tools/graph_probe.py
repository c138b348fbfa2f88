#!/usr/bin/env python3
"""Probe: does replaying the 2-NFE image loop as a HIP graph beat stream launches?  (A/B inside one process.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import ops  # noqa: E402
from bench import N_IMG, build_flux_engine  # noqa: E402

eng, (x0, t, ctx, pooled, guidance, hp, wp) = build_flux_engine('flux', 'cuda')
sig = [1.0, 0.7619047619, 0.0]
tv = [torch.full((1,), s, device='cuda') for s in sig[:2]]
lat = torch.randn(1, N_IMG, 64, device='cuda')


def one_image():
    x = lat
    for i in range(2):
        out = eng(x.bfloat16(), tv[i], ctx, pooled, guidance, hp, wp)
        x = ops.arcflow_step(x, out.means, out.logweights, out.loggammas, sig[i], sig[i], sig[i + 1])
    return x


def timeit(fn, n=8):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one_image()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    res = one_image()
for r in range(3):
    print(f'stream launches {timeit(one_image):7.2f} ms/image | graph replay {timeit(g.replay):7.2f} ms/image')
