#!/usr/bin/env python3
"""Throughput with 1 vs 2 images in flight on one GPU (two engine contexts bound to the SAME weights, one HIP stream each):
does a second stream fill the under-filled last rounds of the GEMM (216 / 648 / 864 tiles on 256 CUs) and attention
(864 work-groups on 512 slots) launches?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import MMDiTEngine, ops  # noqa: E402
from arcflow_amd.weights import random_packed  # noqa: E402

N_IMG, T = 4096, 512
dev = 'cuda'
P = random_packed('flux', 19, 38, dev, seed=0)
nstream = int(sys.argv[1]) if len(sys.argv) > 1 else 2
engs = []
for i in range(nstream):
    e = MMDiTEngine('flux', 19, 38, device=dev)
    e.bind_packed(P)
    engs.append(e)
g = torch.Generator(device=dev).manual_seed(42)
ctx = (torch.randn(1, T, 4096, generator=g, device=dev) * 0.1).bfloat16()
pooled = (torch.randn(1, 768, generator=g, device=dev) * 0.1).bfloat16()
guid = torch.full((1,), 3.5, device=dev)
sig = [1.0, 0.7619, 0.0]
tv = [torch.full((1,), s, device=dev) for s in sig[:2]]
lat = torch.randn(1, N_IMG, 64, device=dev, generator=g)
streams = [torch.cuda.Stream() for _ in range(nstream)]


def image(eng):
    x = lat
    for i in range(2):
        out = eng(x.bfloat16(), tv[i], ctx, pooled, guid, 64, 64)
        x = ops.arcflow_step(x, out.means, out.logweights, out.loggammas, sig[i], sig[i], sig[i + 1])
    return x


def run(n_images):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n_images):
        s = k % nstream
        with torch.cuda.stream(streams[s]):
            image(engs[s])
    torch.cuda.synchronize()
    return n_images / (time.perf_counter() - t0)


run(2 * nstream)
for _ in range(3):
    print(f'{nstream} stream(s): {run(8):.3f} images/s', flush=True)
