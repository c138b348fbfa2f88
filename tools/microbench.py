#!/usr/bin/env python3
"""Per-kernel timings on one MI355X (HIP events around repeated launches, random data).

  python tools/microbench.py [gemm] [attn] [elem] [forward]
"""
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from arcflow_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def gemm():
    print('--- bf16 linear  C = A.W^T (+bias)')
    shapes = [(4608, 9216, 3072), (4608, 3072, 3072), (4608, 12288, 3072), (4608, 3072, 12288),
              (4608, 21504, 3072), (4608, 3072, 15360), (4096, 1152, 3072), (512, 9216, 3072), (8192, 8192, 8192)]
    for M, N, K in shapes:
        a = torch.randn(M, K, device='cuda').bfloat16()
        w = (torch.randn(N, K, device='cuda') * 0.02).bfloat16()
        b = torch.randn(N, device='cuda').bfloat16()
        out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        ws = ops.stream_k_workspace()
        dt = timeit(lambda: ops.linear(a, w, b, out=out))
        dk = timeit(lambda: ops.linear(a, w, b, out=out, sk_ws=ws))
        ref = timeit(lambda: torch.nn.functional.linear(a, w, b))
        print(f'M={M:5d} N={N:5d} K={K:5d}  plain {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:7.1f} TF | stream-K tail {dk*1e6:8.1f} us {2*M*N*K/dk/1e12:7.1f} TF'
              f'   (torch/hipBLASLt {2*M*N*K/ref/1e12:7.1f} TF)')


def tn():
    print('--- TN product (LoRA weight gradients): C[N1,N2] += X[M,N1]^T Y[M,N2], fp32 out')
    import os
    for M, N1, N2 in [(4608, 3072, 256), (4608, 256, 3072), (4608, 12288, 256), (4608, 256, 12288), (4608, 256, 15360), (512, 3072, 256), (4096, 12288, 256),
                      (18432, 3072, 256), (18432, 256, 3072), (18432, 12288, 256)]:
        x = torch.randn(M, N1, device='cuda').bfloat16()
        y = torch.randn(M, N2, device='cuda').bfloat16()
        out = torch.zeros(N1, N2, device='cuda')
        dt = timeit(lambda: ops.linear_tn_f32out(x, y, out=out, accumulate=True))
        lib = __import__('arcflow_amd')._lib.load()
        ks = lib.afx_linear_tn_ws_bytes(M, N1, N2) // (4 * N1 * N2)
        print(f'M={M:5d} N1={N1:5d} N2={N2:5d}  {dt*1e6:8.1f} us {2*M*N1*N2/dt/1e12:7.1f} TF   token split {ks}')
    print('--- rank-256 forward products (128x128 tiles / split-K): t = x A^T')
    for M, K in [(4608, 3072), (4608, 12288), (4608, 15360), (4096, 3072), (512, 3072), (18432, 3072), (18432, 12288)]:
        x = torch.randn(M, K, device='cuda').bfloat16()
        w = (torch.randn(256, K, device='cuda') * 0.02).bfloat16()
        out = torch.empty(M, 256, device='cuda', dtype=torch.bfloat16)
        dt = timeit(lambda: ops.linear(x, w, out=out))
        dk = timeit(lambda: ops.linear_splitk(x, w, out=out, split_k=8))
        print(f'M={M:5d} N=  256 K={K:5d}  plain {dt*1e6:8.1f} us {2*M*256*K/dt/1e12:7.1f} TF | split-K 8 {dk*1e6:8.1f} us {2*M*256*K/dk/1e12:7.1f} TF')


def gemm8():
    print('--- fp8 (e4m3, row-wise scales) linear vs the bf16 kernel')
    for M, N, K in [(4608, 9216, 3072), (4608, 21504, 3072), (4608, 3072, 15360), (4608, 12288, 3072), (4608, 3072, 3072), (4608, 3072, 12288),
                    (9216, 3072, 3072), (9216, 9216, 3072), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device='cuda').bfloat16()
        w = (torch.randn(N, K, device='cuda') * 0.02).bfloat16()
        out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        aq, asc = ops.quant_rows_fp8(a)
        wq, wsc = ops.quant_rows_fp8(w)
        d8 = timeit(lambda: ops.linear_fp8(aq, asc, wq, wsc, out=out))
        dq = timeit(lambda: ops.quant_rows_fp8(a))
        d16 = timeit(lambda: ops.linear(a, w, out=out))
        ref = a.float() @ w.float().t()
        err = ((ops.linear_fp8(aq, asc, wq, wsc).float() - ref).norm() / ref.norm()).item()
        lt = ''
        try:        # hipBLASLt's fp8 GEMM through torch._scaled_mm (test-side reference only; tensor-wise scales: the form every build supports)
            one = torch.ones((), device='cuda')
            a8, w8 = aq.view(torch.float8_e4m3fn), wq.view(torch.float8_e4m3fn)
            dl = timeit(lambda: torch._scaled_mm(a8, w8.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16))
            lt = f' | torch._scaled_mm (hipBLASLt) {dl*1e6:8.1f} us {2*M*N*K/dl/1e12:7.1f} TF'
        except Exception as e:      # noqa: BLE001
            lt = f' | torch._scaled_mm unavailable ({type(e).__name__})'
        mx = ''
        if K % 512 == 0:
            am, ax = ops.quant_rows_mx8(a)
            ones = torch.ones(M, device='cuda')
            dm = timeit(lambda: ops.linear_fp8_mx(am, ax, wq, wsc, out=out, a_scale=ones))
            dqm = timeit(lambda: ops.quant_rows_mx8(a))
            em = ((ops.linear_fp8_mx(am, ax, wq, wsc).float() - ref).norm() / ref.norm()).item()
            mx = f' | block-scaled {dm*1e6:8.1f} us {2*M*N*K/dm/1e12:7.1f} TF quant {dqm*1e6:6.1f} us err {em:.2e}'
        print(f'M={M:5d} N={N:5d} K={K:5d}  fp8 {d8*1e6:8.1f} us {2*M*N*K/d8/1e12:7.1f} TF | quant(A) {dq*1e6:6.1f} us | bf16 {d16*1e6:8.1f} us {2*M*N*K/d16/1e12:7.1f} TF | rel err {err:.2e}{mx}{lt}')


def attn():
    print('--- joint attention, d=128')
    for B, S, H in [(1, 4608, 24), (1, 4224, 24), (1, 1024, 24)]:
        q, k, v = (torch.randn(B, S, H, 128, device='cuda').bfloat16() for _ in range(3))
        dt = timeit(lambda: ops.attention(q, k, v), iters=10)
        ref = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)), iters=10)
        fl = 4.0 * B * H * S * S * 128
        print(f'B={B} S={S} H={H}  {dt*1e6:9.1f} us  {fl/dt/1e12:7.1f} TF   (torch SDPA {fl/ref/1e12:7.1f} TF)')


def attn1():
    q, k, v = (torch.randn(1, 4608, 24, 128, device='cuda').bfloat16() for _ in range(3))
    dt = timeit(lambda: ops.attention(q, k, v), iters=10)
    print(f'attn S=4608: {dt*1e6:.1f} us {4.0*24*4608*4608*128/dt/1e12:.1f} TF')


def gemm1():
    M, N, K = 4608, 21504, 3072
    a = torch.randn(M, K, device='cuda').bfloat16(); w = (torch.randn(N, K, device='cuda') * 0.02).bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    dt = timeit(lambda: ops.linear(a, w, None, out=out))
    print(f'gemm {M}x{N}x{K}: {dt*1e6:.1f} us {2*M*N*K/dt/1e12:.1f} TF')


def elem():
    print('--- HBM-bound kernels')
    R, D = 4608, 3072
    x = torch.randn(R, D, device='cuda').bfloat16()
    sc, sh = torch.randn(1, D, device='cuda'), torch.randn(1, D, device='cuda')
    dt = timeit(lambda: ops.norm_modulate(x, sc, sh))
    print(f'norm_modulate {R}x{D}: {dt*1e6:8.1f} us  {2*R*D*2/dt/1e9:8.1f} GB/s')
    qk = torch.randn(1, R, 24, 128, device='cuda').bfloat16()
    w = torch.ones(128, device='cuda')
    cos, sin = torch.randn(R, 64, device='cuda'), torch.randn(R, 64, device='cuda')
    dt = timeit(lambda: ops.qk_norm_rope_(qk, w, w, cos, sin, 512))
    print(f'qk_norm_rope {R}x24x128: {dt*1e6:8.1f} us  {2*R*D*2/dt/1e9:8.1f} GB/s')
    N = 1056768
    xw = torch.randn(1, D, device='cuda')
    W = torch.randn(N // 8, D, device='cuda').bfloat16()
    dt = timeit(lambda: ops.gemv(xw, W, None))
    print(f'gemv {N//8}x{D}: {dt*1e6:8.1f} us  {W.numel()*2/dt/1e9:8.1f} GB/s')
    B, Nt, K, ch, pp = 1, 4096, 16, 64, 4
    xs = torch.randn(B, Nt, ch, device='cuda')
    m = torch.randn(B, Nt, K, ch, device='cuda').bfloat16()
    lw = torch.log_softmax(torch.randn(B, Nt, K, pp, device='cuda'), 2).bfloat16()
    lg = torch.randn(B, Nt, K - 1, pp, device='cuda').bfloat16()
    out = torch.empty_like(xs)
    dt = timeit(lambda: ops.arcflow_step(xs, m, lw, lg, 1.0, 1.0, 0.76, out=out), iters=50)
    byts = xs.numel() * 8 + (m.numel() + lw.numel() + lg.numel()) * 2
    print(f'arcflow_step bf16 mix 4096 tok: {dt*1e6:8.1f} us  {byts/dt/1e9:8.1f} GB/s')
    mf, lwf, lgf = m.float(), lw.float(), lg.float()
    dt = timeit(lambda: ops.arcflow_step(xs, mf, lwf, lgf, 1.0, 1.0, 0.76, out=out), iters=50)
    byts = xs.numel() * 8 + (m.numel() + lw.numel() + lg.numel()) * 4
    print(f'arcflow_step f32 mix 4096 tok: {dt*1e6:8.1f} us  {byts/dt/1e9:8.1f} GB/s')


def forward():
    print('--- FLUX-12B architecture forward, random weights, 1024^2 (N=4096, T=512)')
    from bench import build_flux_engine
    eng, inputs = build_flux_engine()
    fn = lambda: eng(*inputs)  # noqa: E731
    t0 = time.time()
    fn(); torch.cuda.synchronize()
    print(f'first call {time.time()-t0:.2f}s')
    dt = timeit(fn, iters=5, warm=2)
    print(f'forward {dt*1e3:8.2f} ms  -> {74.41e12/dt/1e12:7.1f} TF effective, {1/(2*dt):6.2f} img/s (2 NFE, DiT only)')


if __name__ == '__main__':
    what = sys.argv[1:] or ['gemm', 'attn', 'elem', 'forward']
    for w in what:
        globals()[w]()
