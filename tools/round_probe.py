#!/usr/bin/env python3
"""Is a launch's time set by its ROUNDS (work-groups / 256 CUs, rounded up) or by its WORK (the power cap: energy per flop)?
Attention at S = 4608 over the head count (18 work-groups per head), the 256x256-tile GEMM at M = 4608, K = 3072 over N (18 tiles per
256 columns).  If time follows ceil(rounds) the tail of an under-filled round is worth recovering (KV-split, stream-K); if it follows
the work, idle CUs only lend their power to the busy ones.   usage: python tools/round_probe.py [--reps 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from arcflow_amd import ops  # noqa: E402


def timed(fn, reps):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=30)
    a = ap.parse_args()
    S = 4608
    g = torch.Generator(device='cuda').manual_seed(1)
    print('# attention S=4608: heads, work-groups, rounds, us, us per unit of work (us / rounds), TF/s')
    for rnd in range(2):
        for H in (8, 12, 14, 16, 20, 24, 26, 28, 32, 40, 42):
            q, k, v = (torch.randn(1, S, H, 128, generator=g, device='cuda').bfloat16() for _ in range(3))
            o = torch.empty(1, S, H * 128, device='cuda', dtype=torch.bfloat16)
            vt = ops.v_transpose(v) if hasattr(ops, 'v_transpose') else None
            if vt is not None and hasattr(ops, 'attention_vt'):
                fn = lambda: ops.attention_vt(q, k, vt, out=o)
            else:
                fn = lambda: ops.attention(q, k, v)
            us = timed(fn, a.reps)
            wgs = H * 18
            print(f'attn H={H:3d} wgs={wgs:4d} rounds={wgs / 256:5.2f} {us:8.1f} us  {us / (wgs / 256):7.1f} us/round-of-work  '
                  f'{4 * H * S * S * 128 / us / 1e6:7.0f} TF/s', flush=True)
    print('# GEMM 256x256 tiles, M=4608 K=3072: N, tiles, rounds, us, us per unit of work, TF/s')
    ops.set_gemm_mode(3, 1)
    M, K = 4608, 3072
    a_ = torch.randn(M, K, generator=g, device='cuda').bfloat16()
    for rnd in range(2):
        for N in (3584, 4096, 5120, 6144, 7168, 7424, 8192, 9216, 10240, 10752, 11008, 12288, 14336):
            w = (torch.randn(N, K, generator=g, device='cuda') * 0.02).bfloat16()
            b = torch.randn(N, generator=g, device='cuda').bfloat16()
            out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
            us = timed(lambda: ops.linear(a_, w, b, out=out), a.reps)
            tiles = 18 * ((N + 255) // 256)
            print(f'gemm N={N:6d} tiles={tiles:4d} rounds={tiles / 256:5.2f} {us:8.1f} us  {us / (tiles / 256):7.1f} us/round-of-work  '
                  f'{2 * M * N * K / us / 1e6:7.0f} TF/s', flush=True)
    ops.set_gemm_mode(3, 0)


if __name__ == '__main__':
    main()
