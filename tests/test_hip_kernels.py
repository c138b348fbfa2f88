"""GPU parity tests of the HIP kernels, through the C ABI (ctypes), against the CPU oracle / fp32 torch.

bf16 kernels: tolerance stated per test (inputs are bf16-rounded, the oracle computes in fp32).
fp32 ArcFlow step: rel 1e-5.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from arcflow_amd import ops as _ops, _lib
    _lib.load()
    return _ops


def dev():
    return torch.device('cuda:0')


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp(min=1e-12)).item()


def bf(t):
    return t.to(torch.bfloat16)


# ------------------------------------------------------------------------------------------ GEMM
# (impl, tile) of afx_gemm_set_mode: the one-wave-per-SIMD kernel with the tile shape picked per launch / forced to 256x256 /
# 288x192 / 320x192 / 128x128 / 256x224 / 224x256, and the 8-phase 256x256 kernel.  Every mode must give the same results on the same inputs.
GEMM_MODES = [(3, 0), (3, 1), (3, 2), (3, 3), (3, 4), (3, 5), (3, 6), (2, 0)]
GEMM_MODE_IDS = ['auto', 'v3-256x256', 'v3-288x192', 'v3-320x192', 'v3-128x128', 'v3-256x224', 'v3-224x256', '8phase']


@pytest.fixture
def gemm_mode(request, ops):
    impl, tile = request.param
    ops.set_gemm_mode(impl, tile)
    yield request.param
    ops.set_gemm_mode(3, 0)


all_gemm_modes = pytest.mark.parametrize('gemm_mode', GEMM_MODES, ids=GEMM_MODE_IDS, indirect=True)


@all_gemm_modes
@pytest.mark.parametrize('M,N,K', [(256, 256, 64), (300, 264, 128), (77, 3072, 4096), (1000, 520, 3072), (513, 1152, 192), (640, 576, 64),
                                   (289, 200, 192)])
def test_linear_plain(ops, gemm_mode, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a = bf(torch.randn(M, K, generator=g)).to(dev())
    w = bf(torch.randn(N, K, generator=g) * 0.05).to(dev())
    b = bf(torch.randn(N, generator=g)).to(dev())
    out = ops.linear(a, w, b)
    ref = a.float() @ w.float().T + b.float()
    assert rel_l2(out, ref) < 4e-3            # bf16 output rounding only (fp32 accumulate)
    assert (out.float() - ref).abs().max().item() < 0.05 * ref.abs().max().item()


@all_gemm_modes
def test_linear_identity_asymmetric(ops, gemm_mode):
    """A = I against an asymmetric W catches row/col swaps in the MFMA -> C mapping exactly."""
    n = 512
    a = torch.eye(n, dtype=torch.bfloat16, device=dev())
    w = bf(torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125).to(dev())   # exact in bf16
    out = ops.linear(a, w, None)
    assert torch.equal(out.float(), w.float().T)


@all_gemm_modes
def test_linear_strided_gelu_cols(ops, gemm_mode):
    g = torch.Generator().manual_seed(5)
    M, K, N = 333, 256, 512
    big = bf(torch.randn(M, 3 * K, generator=g)).to(dev())
    a = big[:, K:2 * K]                                   # lda = 3K
    w = bf(torch.randn(N, K, generator=g) * 0.06).to(dev())
    b = bf(torch.randn(N, generator=g)).to(dev())
    outbuf = torch.zeros(M, N + 64, dtype=torch.bfloat16, device=dev())
    out = ops.linear(a, w, b, epilogue='gelu', gelu_col0=256, out=outbuf[:, :N])
    y = a.float() @ w.float().T + b.float()
    ref = torch.cat([y[:, :256], torch.nn.functional.gelu(y[:, 256:], approximate='tanh')], dim=1)
    assert rel_l2(out, ref) < 4e-3
    assert outbuf[:, N:].abs().max().item() == 0           # nothing written past N


@all_gemm_modes
def test_linear_gate_residual_inplace(ops, gemm_mode):
    g = torch.Generator().manual_seed(6)
    B, S, K, N = 2, 150, 512, 256
    a = bf(torch.randn(B * S, K, generator=g)).to(dev())
    w = bf(torch.randn(N, K, generator=g) * 0.05).to(dev())
    b = bf(torch.randn(N, generator=g)).to(dev())
    gate = torch.randn(B, N, generator=g).to(dev())
    x = bf(torch.randn(B * S, N, generator=g)).to(dev())
    ref = x.float() + gate.repeat_interleave(S, 0) * (a.float() @ w.float().T + b.float())
    out = ops.linear(a, w, b, epilogue='gate_res', gate=gate, residual=x, rows_per_batch=S, out=x)
    assert out.data_ptr() == x.data_ptr()
    assert rel_l2(out, ref) < 4e-3


# ------------------------------------------------------------------------------------------ attention
# afx_attn_set_impl: 0 = the one-wave-per-SIMD kernel (afx_attn3.hip; any S > 64 -- shorter sequences fall back to the 4-wave kernel),
# 1 = the 4-wave kernel always.  Both must match fp32 softmax on the same bf16 inputs.
@pytest.fixture(params=[0, 1], ids=['v3-wave64q', '4wave'])
def attn_impl(request, ops):
    ops.set_attn_impl(request.param)
    yield request.param
    ops.set_attn_impl(0)


def _sdpa_ref(q, k, v):
    B, S, H, _ = q.shape
    return torch.nn.functional.scaled_dot_product_attention(
        q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)).transpose(1, 2).reshape(B, S, H * 128)


@pytest.mark.parametrize('B,S,H', [(1, 64, 1), (2, 200, 2), (1, 333, 3), (1, 1024, 2), (1, 129, 1), (1, 128, 1), (1, 192, 2), (2, 320, 3),
                                   (1, 576, 9), (1, 2048, 4), (1, 65, 1), (1, 191, 2), (2, 1101, 3), (1, 767, 1)])
def test_attention(ops, attn_impl, B, S, H):
    """S = 128 (two KV tiles: prologue + last-tile code only), 192 (one loop iteration), 320 / 576 (partly filled 256-query blocks,
    2-4 loop iterations: every ring slot), H = 9 (heads -> XCD map with an incomplete last group), ragged S (the last KV tile's
    keys past the end start from -inf through the key mask: 65 = one real key in the last tile, 191 = one masked key, 333, 767, 1101)."""
    g = torch.Generator().manual_seed(S)
    q = bf(torch.randn(B, S, H, 128, generator=g)).to(dev())
    k = bf(torch.randn(B, S, H, 128, generator=g)).to(dev())
    v = bf(torch.randn(B, S, H, 128, generator=g)).to(dev())
    out = ops.attention(q, k, v)
    ref = _sdpa_ref(q, k, v)
    assert rel_l2(out, ref) < 1.2e-2           # P is rounded to bf16 before P.V, output bf16
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize('S', [320, 1024])
def test_attention_spiked_rows(ops, attn_impl, S):
    """One key dominating a query row late in the sequence forces the online-softmax rescale (the cold path of the one-wave-per-SIMD
    kernel: accumulator-file reads / writes with hand-placed wait states) in both slabs of a wave and at an early and a late tile."""
    g = torch.Generator().manual_seed(3)
    B, H = 1, 2
    q = torch.randn(B, S, H, 128, generator=g)
    k = torch.randn(B, S, H, 128, generator=g)
    v = torch.randn(B, S, H, 128, generator=g)
    k[0, S - 20, 0] = q[0, 7, 0] * 3.0            # slab A of wave 0, last tile
    k[0, 10, 0] = q[0, 100, 0] * 3.0              # slab B of wave 1, first tile
    k[0, 200, 1] = q[0, 40, 1] * 4.0              # slab B of wave 0, a middle tile
    k[0, 130, 1] = q[0, S - 1, 1] * 4.0
    q, k, v = (bf(t).to(dev()) for t in (q, k, v))
    out = ops.attention(q, k, v)
    ref = _sdpa_ref(q, k, v)
    assert (out.float() - ref).abs().max().item() < 0.06
    assert rel_l2(out, ref) < 1.2e-2


def test_attention_growing_scores_every_tile(ops, attn_impl):
    """Keys whose scores grow along the sequence: the running max of EVERY row outgrows the deferral threshold (2^5) again and again,
    so the rescale branch runs many times per work-group and must leave l, O and the later P on one scale."""
    g = torch.Generator().manual_seed(11)
    B, S, H = 1, 1024, 1
    q = torch.randn(B, S, H, 128, generator=g)
    base = torch.randn(128, generator=g)
    base = base / base.norm()
    q = q + 6.0 * base                                            # every query has a large component along `base`
    k = torch.randn(B, S, H, 128, generator=g) + (torch.arange(S).float() / S * 60.0).view(1, S, 1, 1) * base
    v = torch.randn(B, S, H, 128, generator=g)
    q, k, v = (bf(t).to(dev()) for t in (q, k, v))
    out = ops.attention(q, k, v)
    ref = _sdpa_ref(q, k, v)
    assert torch.isfinite(out.float()).all()
    assert rel_l2(out, ref) < 1.2e-2


def test_attention_kernels_bit_identical(ops):
    """Same mathematics, same operation order per query: the two kernels agree bit for bit (S % 64 == 0)."""
    g = torch.Generator().manual_seed(5)
    q, k, v = (bf(torch.randn(1, 768, 3, 128, generator=g)).to(dev()) for _ in range(3))
    ops.set_attn_impl(1)
    a = ops.attention(q, k, v)
    ops.set_attn_impl(0)
    b = ops.attention(q, k, v)
    assert (a.float() - b.float()).abs().max().item() < 4e-3      # row sums are accumulated in a different order (two partial sums)


def test_attention_lse_matches_between_kernels(ops):
    g = torch.Generator().manual_seed(6)
    q, k, v = (bf(torch.randn(1, 384, 2, 128, generator=g)).to(dev()) for _ in range(3))
    ops.set_attn_impl(1)
    o1, l1 = ops.attention_fwd_lse(q, k, v)
    ops.set_attn_impl(0)
    o0, l0 = ops.attention_fwd_lse(q, k, v)
    assert (l0 - l1).abs().max().item() < 2e-3
    s = torch.einsum('bqhd,bkhd->bhqk', q.float(), k.float()) * (128 ** -0.5) * 1.4426950408889634
    ref = torch.logsumexp(s * 0.6931471805599453, dim=-1) * 1.4426950408889634          # log2-domain log-sum-exp
    assert (l0 - ref).abs().max().item() < 2e-2


@pytest.mark.parametrize('B,S,H', [(1, 4608, 24), (1, 4173, 24), (1, 4224, 24), (2, 2300, 5), (1, 1024, 2), (1, 4608, 8), (3, 4608, 24)])
def test_attention_kv_split_of_the_last_round(ops, B, S, H):
    """Round 5: the 256-query blocks of an under-filled last round are cut into one run of key tiles per CU (a work-group then runs one or two
    segments, a segment writes a normalised partial + its log-sum-exp, attention_combine_kernel merges them).  Against the same kernel on
    its plain grid (impl 3) and against fp32 softmax: FLUX / Qwen / ragged shapes, short sequences (runs of 8 tiles), several samples,
    O written over Q in place as the engine does, and the log-sum-exp the training forward keeps."""
    g = torch.Generator(device='cuda').manual_seed(S + H)
    q, k, v = (torch.randn(B, S, H, 128, generator=g, device='cuda').bfloat16() for _ in range(3))
    ref = _sdpa_ref(q, k, v)
    outs = {}
    for impl in (3, 0):
        ops.set_attn_impl(impl)
        qo = q.clone().reshape(B * S, H * 128)
        lse = ops.attention_fwd_lse_2d(qo, k.reshape(B * S, H * 128), v.reshape(B * S, H * 128), qo, B, S, H)      # O over Q
        outs[impl] = (qo.reshape(B, S, H * 128).float(), lse[:, :, :S].clone())
    ops.set_attn_impl(0)
    for impl in (3, 0):
        assert rel_l2(outs[impl][0], ref) < 1.2e-2
        assert torch.isfinite(outs[impl][0]).all()
    assert rel_l2(outs[0][0], outs[3][0]) < 4e-3                  # a merged row is rounded to bf16 twice
    assert (outs[0][0] - outs[3][0]).abs().max().item() < 0.03
    assert (outs[0][1] - outs[3][1]).abs().max().item() < 2e-3


def test_attention_plan_cache_is_bounded_and_plans_survive_eviction(ops):
    """Prompt lengths vary batch to batch in training (S = T + N): the per-(B, H, S, stream) work tables of the balanced grid live in a cache of at most 16 plans,
    the least recently used one is freed when a new shape arrives, a new plan is uploaded on the launch stream.  20 distinct lengths, then the first one again (its plan was
    evicted and is rebuilt): every result equals the plain grid's up to the double rounding of merged rows, and device memory does not grow with the number of shapes."""
    H = 24
    g = torch.Generator(device='cuda').manual_seed(5)
    lengths = [4608 - 64 * i for i in range(20)] + [4608, 4544]
    qkv = {S: tuple(torch.randn(1, S, H, 128, generator=g, device='cuda').bfloat16() for _ in range(3)) for S in set(lengths)}
    ops.set_attn_impl(3)
    ref = {S: ops.attention(*qkv[S]).float() for S in set(lengths)}
    ops.set_attn_impl(0)
    torch.cuda.synchronize()
    free0 = None
    for n, S in enumerate(lengths):
        out = ops.attention(*qkv[S]).float()
        assert torch.isfinite(out).all()
        assert rel_l2(out, ref[S]) < 4e-3, S
        if n == 17:                                   # the cache is full (16 plans): from here on every new shape must free an old plan
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)        # a plan is <= ~26 MB; an unbounded cache would have grown by four more


def test_attention_hand_over_fallback_when_partials_never_arrive(ops, monkeypatch):
    """HIP promises no dispatch order: a long part that does not see its block's partials published must still produce the right rows.  With
    AFX_ATTN_HANDOVER=lost the long parts act as if no flag were ever raised and compute their whole key range from zero (the short ends' work is
    then wasted, never wrong)."""
    g = torch.Generator(device='cuda').manual_seed(99)
    B, S, H = 1, 4608, 24
    q, k, v = (torch.randn(B, S, H, 128, generator=g, device='cuda').bfloat16() for _ in range(3))
    ops.set_attn_impl(3)
    plain = ops.attention(q, k, v).float()
    ops.set_attn_impl(0)
    monkeypatch.setenv('AFX_ATTN_HANDOVER', 'lost')
    lost = ops.attention(q, k, v).float()
    monkeypatch.delenv('AFX_ATTN_HANDOVER')
    torch.cuda.synchronize()
    assert torch.equal(lost, plain)            # whole key range, same order of operations: the plain grid's bits
    handed = ops.attention(q, k, v).float()
    assert not torch.equal(handed, plain) and rel_l2(handed, plain) < 4e-3


# ------------------------------------------------------------------------------------------ races / determinism (tools/race_probe*.py)
def _count_nonidentical(fn, reps, junk):
    ref = fn().clone()
    bad = 0
    for i in range(reps):
        if i % 3 == 0:
            junk.normal_()                     # 256 MB of writes: evicts L2 / Infinity Cache, shifts the timing of the next launch
        bad += int(not torch.equal(fn(), ref))
    return bad


@pytest.mark.parametrize('S', [4608, 4173])
def test_attention_race_probe(ops, S):
    """100+ launches on fixed inputs with cache-disturbing work in between: every output bit-identical.  This is the probe that found
    the compiler-dropped DMA wait of round 2: the one-wave-per-SIMD kernel (hand-counted vmcnt / lgkmcnt waits, LDS rings)
    at S = 4608 and on the ragged S = 4173 (clamped last-tile DMA, key mask)."""
    g = torch.Generator(device='cuda').manual_seed(0)
    q, k, v = (torch.randn(1, S, 24, 128, generator=g, device='cuda').bfloat16() for _ in range(3))
    junk = torch.empty(64 << 20, dtype=torch.float32, device='cuda')
    ops.set_attn_impl(0)
    assert _count_nonidentical(lambda: ops.attention(q, k, v), 120, junk) == 0


@pytest.mark.parametrize('S', [4608, 4173])
def test_attention_backward_race_probe(ops, S):
    """The generated backward streams (afx_attn_bwd3.hip: hand-counted vmcnt / lgkmcnt waits, an 8-slot LDS ring, one barrier per phase, the dK / dV and the dQ
    work-groups in one grid): 60 calls on fixed inputs with cache-disturbing work in between, dq | dk | dv bit-identical every time -- at S = 4608 and on a ragged
    S = 4173 (clamped last-tile DMA rows, padded L | -delta side array, the dQ stream's cold key mask)."""
    g = torch.Generator(device='cuda').manual_seed(1)
    q, k, v, do = (torch.randn(1, S, 24, 128, generator=g, device='cuda').bfloat16() for _ in range(4))
    o, lse = ops.attention_fwd_lse(q, k, v)
    o = o.reshape(1, S, 24, 128)
    junk = torch.empty(64 << 20, dtype=torch.float32, device='cuda')
    assert _count_nonidentical(lambda: torch.cat([t.reshape(-1) for t in ops.attention_bwd(q, k, v, o, do, lse)]), 60, junk) == 0


@pytest.mark.parametrize('M,N,K', [(4608, 3072, 3072), (4608, 9216, 3072), (4608, 3072, 15360), (512, 9216, 3072)])
def test_gemm_race_probe(ops, M, N, K):
    g = torch.Generator(device='cuda').manual_seed(0)
    a = torch.randn(M, K, generator=g, device='cuda').bfloat16()
    w = (torch.randn(N, K, generator=g, device='cuda') * 0.02).bfloat16()
    b = torch.randn(N, generator=g, device='cuda').bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    junk = torch.empty(64 << 20, dtype=torch.float32, device='cuda')
    assert _count_nonidentical(lambda: ops.linear(a, w, b, out=out), 100, junk) == 0


@pytest.mark.parametrize('M,N,K,mx', [(4224, 3072, 3072, False), (4608, 9216, 3072, False), (4224, 12288, 3072, True), (4173, 3072, 12288, True)])
def test_fp8_gemm_race_probe_both_tile_shapes(ops, M, N, K, mx, monkeypatch):
    """The one-wave-per-SIMD fp8 kernel on its two tile shapes (256x256 and, round 5, 224x256: the launcher picks per launch; these shapes take 224x256): 60 launches
    on fixed inputs with cache-disturbing work in between, every output bit-identical, and the result within the fp8 tolerance of the fp32 product."""
    g = torch.Generator(device='cuda').manual_seed(3)
    a = torch.randn(M, K, generator=g, device='cuda').bfloat16()
    w = (torch.randn(N, K, generator=g, device='cuda') * 0.02).bfloat16()
    b = torch.randn(N, generator=g, device='cuda').bfloat16()
    wq, ws = ops.quant_rows_fp8(w)
    junk = torch.empty(64 << 20, dtype=torch.float32, device='cuda')
    if mx:
        aq, amx = ops.quant_rows_mx8(a)
        fn = lambda: ops.linear_fp8_mx(aq, amx, wq, ws, b)      # noqa: E731
    else:
        aq, asc = ops.quant_rows_fp8(a)
        fn = lambda: ops.linear_fp8(aq, asc, wq, ws, b)         # noqa: E731
    assert _count_nonidentical(fn, 60, junk) == 0
    auto = fn().clone()
    ref = (a.float() @ w.float().t() + b.float())
    assert rel_l2(auto.float(), ref) < 6e-2


# ------------------------------------------------------------------------------------------ norms / rope / gemv
@pytest.mark.parametrize('D', [256, 3072, 3584])
def test_norm_modulate(ops, D):
    g = torch.Generator().manual_seed(D)
    B, S = 2, 37
    x = bf(torch.randn(B * S, D, generator=g) * 2 + 0.3).to(dev())
    sc = torch.randn(B, D, generator=g).to(dev())
    sh = torch.randn(B, D, generator=g).to(dev())
    out = ops.norm_modulate(x, sc, sh, rows_per_batch=S)
    xf = x.float()
    ref = torch.nn.functional.layer_norm(xf, (D,), eps=1e-6) * (1 + sc.repeat_interleave(S, 0)) + sh.repeat_interleave(S, 0)
    assert rel_l2(out, ref) < 4e-3
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(dev())
    out = ops.norm_modulate(x, w, None, rms=True)
    ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * w
    assert rel_l2(out, ref) < 4e-3


def test_qk_norm_rope(ops):
    from oracle import dit_ref as D
    g = torch.Generator().manual_seed(11)
    B, hp, wp, T, H = 2, 5, 6, 9, 3
    S = T + hp * wp
    x = bf(torch.randn(B, S, H, 128, generator=g)).to(dev())
    wt = (1 + 0.1 * torch.randn(128, generator=g))
    wi = (1 + 0.1 * torch.randn(128, generator=g))
    cos, sin = D.flux_rope_tables(hp, wp, T)
    ref_t = D.apply_rope(D.rms_norm(x.float().cpu()[:, :T], wt), cos[:T], sin[:T])
    ref_i = D.apply_rope(D.rms_norm(x.float().cpu()[:, T:], wi), cos[T:], sin[T:])
    ref = torch.cat([ref_t, ref_i], dim=1)
    out = ops.qk_norm_rope_(x.clone(), wt.to(dev()), wi.to(dev()), cos.to(dev()), sin.to(dev()), T)
    assert rel_l2(out, ref) < 4e-3
    # Qwen tables through the same kernel
    from arcflow_amd import rope
    ia, ta = D.qwen_rope_angles(hp, wp, T)
    c2, s2 = rope.qwen_tables(hp, wp, T)
    assert torch.allclose(c2, torch.cat([torch.cos(ta), torch.cos(ia)]), atol=1e-6)
    assert torch.allclose(s2, torch.cat([torch.sin(ta), torch.sin(ia)]), atol=1e-6)


def test_gemv(ops):
    g = torch.Generator().manual_seed(12)
    B, N, K = 3, 1030, 3072
    x = torch.randn(B, K, generator=g).to(dev())
    w = bf(torch.randn(N, K, generator=g) * 0.02).to(dev())
    b = bf(torch.randn(N, generator=g)).to(dev())
    ref = x @ w.float().T + b.float()
    out = ops.gemv(x, w, b)
    assert rel_l2(out, ref) < 1e-5
    out2 = ops.gemv(x, w, b, act='silu', out=out.clone(), accumulate=True)
    assert rel_l2(out2, ref + torch.nn.functional.silu(ref)) < 1e-5


# ------------------------------------------------------------------------------------------ ArcFlow step
def test_arcflow_step_golden(ops, golden):
    """The HIP step in the token layout == the reference's unpack -> policy -> integrate -> repack."""
    g = golden('g5_layouts')
    x, m, lw, lg = (torch.from_numpy(g[k]).to(dev()) for k in ('x_tok', 'means_tok', 'logw_tok', 'logg_tok'))
    out = ops.arcflow_step(x, m, lw, lg, 1.0, 1.0, float(np.float32(761.9047761) / 1000))
    ref = torch.from_numpy(g['x_end_tok'])
    assert torch.allclose(out.cpu(), ref, rtol=1e-5, atol=1e-5)


def test_arcflow_step_full_size_golden(ops, golden_full_size_inputs):
    """One FULL-SIZE step (4096 tokens = latent [1, 16, 128, 128], K = 16, both steps of the 2-NFE schedule) against fixture G9 = the reference's own
    unpack -> policy -> momentum_integration -> repack on the same seeded draw (arcflux_pipeline.py:482-510; SURVEY 8c "plus one full 128 x 128")."""
    from test_oracle_golden import _check_g9
    inp, g = golden_full_size_inputs
    _check_g9(lambda x, m, lw, lg, s0, s1: ops.arcflow_step(x.to(dev()), m.to(dev()), lw.to(dev()), lg.to(dev()), s0, s0, s1), inp, g,
              rtol=1e-5, atol=1e-5)


def test_arcflow_step_vs_oracle_edge_cases(ops):
    from oracle import arcflow_ref as R
    g = torch.Generator().manual_seed(21)
    B, N, K, ch, pp = 2, 50, 16, 64, 4
    x = torch.randn(B, N, ch, generator=g)
    m = torch.randn(B, N, K, ch, generator=g)
    lw = torch.log_softmax(torch.randn(B, N, K, pp, generator=g) * 2, dim=2)
    lg = torch.randn(B, N, K - 1, pp, generator=g)
    lg[0, 0] = 0.0            # phi clamp, z == 0 -> +eps branch
    lg[0, 1] = 1e-5
    lg[0, 2] = -1e-5
    lw[1, 3, 5] = float('-inf')   # dropped component (GM dropout)
    for (s0, s1, s2) in [(1.0, 1.0, 0.7619), (0.7619, 0.7619, 0.0), (1.0, 0.9, 0.4), (0.5, 0.5, 0.5)]:
        ref = R.momentum_step_packed(x, m, lw, lg, s0, s1, s2)
        out = ops.arcflow_step(x.to(dev()), m.to(dev()), lw.to(dev()), lg.to(dev()), s0, s1, s2)
        assert torch.allclose(out.cpu(), ref, rtol=1e-5, atol=2e-5), (s0, s1, s2)
    # per-sample sigmas (training form) + bf16 mixture inputs
    sv = [torch.tensor([1.0, 0.8]), torch.tensor([0.9, 0.6]), torch.tensor([0.5, 0.1])]
    out = ops.arcflow_step(x.to(dev()), m.to(dev()), lw.to(dev()), lg.to(dev()), *[s.to(dev()) for s in sv])
    for b in range(B):
        ref = R.momentum_step_packed(x[b:b + 1], m[b:b + 1], lw[b:b + 1], lg[b:b + 1], *[float(s[b]) for s in sv])
        assert torch.allclose(out[b:b + 1].cpu(), ref, rtol=1e-5, atol=2e-5)
    mb, lwb, lgb = bf(m), bf(lw), bf(lg)
    out = ops.arcflow_step(x.to(dev()), mb.to(dev()), lwb.to(dev()), lgb.to(dev()), 1.0, 1.0, 0.7619)
    ref = R.momentum_step_packed(x, mb.float(), lwb.float(), lgb.float(), 1.0, 1.0, 0.7619)
    assert torch.allclose(out.cpu(), ref, rtol=1e-5, atol=2e-5)
    # velocity
    u = ops.arcflow_velocity(m.to(dev()), lw.to(dev()), lg.to(dev()), 1.0, 0.8)
    hp, wp = 5, 10
    ml, lwl, lgl = R.unpack_mixture(m, lw, lg, hp, wp)
    uref = R.pack_latents(R.policy_velocity(ml, lwl, lgl, 1.0, 0.8))
    assert torch.allclose(u.cpu(), uref, rtol=1e-5, atol=2e-5)


def test_arcflow_step_full_size_properties(ops):
    """1024^2 sizes: (i) a mixture with all rates -> 0 and equal means is an Euler step; (ii) steps compose
    exactly when sigma_src is kept (semigroup of the closed-form transport)."""
    g = torch.Generator().manual_seed(31)
    B, N, K, ch, pp = 1, 4096, 16, 64, 4
    x = torch.randn(B, N, ch, generator=g).to(dev())
    m1 = torch.randn(B, N, 1, ch, generator=g).expand(B, N, K, ch).contiguous().to(dev())
    lw = torch.log_softmax(torch.randn(B, N, K, pp, generator=g), dim=2).to(dev())
    z = torch.zeros(B, N, K - 1, pp, device=dev())
    out = ops.arcflow_step(x, m1, lw, z, 1.0, 1.0, 0.25)
    phi_eps = float(np.expm1(np.float32(1e-4)) / np.float32(1e-4))
    assert torch.allclose(out, x - 0.75 * m1[:, :, 0], rtol=2e-4, atol=2e-4)
    m = torch.randn(B, N, K, ch, generator=g).to(dev())
    lg = (torch.randn(B, N, K - 1, pp, generator=g)).to(dev())
    a = ops.arcflow_step(x, m, lw, lg, 1.0, 1.0, 0.3)
    mid = ops.arcflow_step(x, m, lw, lg, 1.0, 1.0, 0.7)
    b = ops.arcflow_step(mid, m, lw, lg, 1.0, 0.7, 0.3)
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-4)


@all_gemm_modes
@pytest.mark.parametrize('M,N,K', [(4608, 3072, 3072), (4096, 1152, 3072), (512, 9216, 3072), (4608, 3072, 15360), (2304, 21504, 3072)])
def test_linear_full_size_vs_device_reference(ops, gemm_mode, M, N, K):
    """Full-size shapes of the FLUX forward, checked on-device against torch's (hipBLASLt) bf16 linear,
    three launches each so a rare pipeline race (DMA landing late / restaged early) shows up."""
    g = torch.Generator(device='cuda').manual_seed(K + N)
    a = torch.randn(M, K, generator=g, device='cuda').bfloat16()
    w = (torch.randn(N, K, generator=g, device='cuda') * 0.03).bfloat16()
    b = torch.randn(N, generator=g, device='cuda').bfloat16()
    ref = torch.nn.functional.linear(a.float(), w.float(), b.float())
    for _ in range(3):
        out = ops.linear(a, w, b)
        assert rel_l2(out, ref) < 4e-3
        assert (out.float() - ref).abs().max().item() < 0.02 * ref.abs().max().item() + 0.05


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,K,split', [(162, 3584, 18944, 0), (512, 4096, 10240, 0), (77, 768, 3072, 3), (300, 520, 1024, 16)])
def test_linear_splitk_matches_plain_gemm(M, N, K, split):
    from arcflow_amd import ops
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).bfloat16().cuda()
    b = torch.randn(N, generator=g).bfloat16().cuda()
    r = torch.randn(M, N, generator=g).bfloat16().cuda()
    got = ops.linear_splitk(a, w, b, r, split_k=split)
    ref = a.float() @ w.float().t() + b.float() + r.float()
    assert ((got.float() - ref).norm() / ref.norm()).item() < 4e-3


@pytest.mark.gpu
def test_fp8_quant_and_linear():
    """Row-wise e4m3 quantisation (vs torch.float8_e4m3fn) and the fp8 MFMA linear with every epilogue."""
    from arcflow_amd import ops
    g = torch.Generator().manual_seed(3)
    M, N, K = 300, 520, 1024
    a = (torch.randn(M, K, generator=g) * 2.0).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    aq, asc = ops.quant_rows_fp8(a)
    wq, wsc = ops.quant_rows_fp8(w)
    # quantiser: scale = absmax / 448, codes = RNE cast of x / scale (saturating) -- the same as torch's e4m3fn cast
    assert torch.allclose(asc, a.float().abs().amax(1) / 448.0, rtol=1e-6)
    ref_codes = (a.float() / asc[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert (aq != ref_codes).float().mean().item() < 1e-3          # ties on the bf16 -> fp32 product may differ in the last bit
    ad = aq.view(torch.float8_e4m3fn).float() * asc[:, None]
    wd = wq.view(torch.float8_e4m3fn).float() * wsc[:, None]
    exact = ad @ wd.t()                                             # what the fp8 GEMM must reproduce exactly (fp32 accumulate)
    b = torch.randn(N, generator=g).bfloat16().cuda()
    r = torch.randn(M, N, generator=g).bfloat16().cuda()
    gate = torch.randn(N, generator=g).cuda()
    rel = lambda x, y: ((x.float() - y).norm() / y.norm()).item()   # noqa: E731
    assert rel(ops.linear_fp8(aq, asc, wq, wsc, b), exact + b.float()) < 4e-3
    assert rel(ops.linear_fp8(aq, asc, wq, wsc, b, epilogue='gelu'), torch.nn.functional.gelu(exact + b.float(), approximate='tanh')) < 5e-3
    assert rel(ops.linear_fp8(aq, asc, wq, wsc, b, epilogue='gate_res', gate=gate, residual=r), r.float() + gate * (exact + b.float())) < 4e-3
    # and the quantisation error itself against the bf16 product stays at the e4m3 level
    assert rel(ops.linear_fp8(aq, asc, wq, wsc), a.float() @ w.float().t()) < 6e-2


@pytest.mark.gpu
@pytest.mark.parametrize('K', [3072, 12288, 15360, 3584])
def test_fp8_row_quantiser_register_kernel(K):
    """K = 512 n (3072 / 12288 / 15360) takes the quantiser that holds the row in registers; the others the two-pass kernel: scales and codes against
    torch's e4m3fn cast, a strided source included."""
    from arcflow_amd import ops
    g = torch.Generator().manual_seed(17)
    M = 515
    big = (torch.randn(M, K + 64, generator=g) * 3.0).bfloat16().cuda()
    a = big[:, :K]                                                  # row stride K + 64
    aq, asc = ops.quant_rows_fp8(a)
    assert torch.allclose(asc, a.float().abs().amax(1) / 448.0, rtol=1e-6)
    ref = (a.float() / asc[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert (aq != ref).float().mean().item() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('K', [256, 1024, 1152, 3072])
def test_fp8_linear_one_wave_kernel(K):
    """Launches with a full round of 256x256 tiles run gemm_kernel_v3f8 (one wave per SIMD, column-split phases): ragged M / N, an even and an
    odd number of K-tiles (the odd one peels a tile), every epilogue -- against the exact fp32 product of the dequantised operands."""
    from arcflow_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N = 4645, 3640                                             # 19 x 15 tiles, both edges ragged
    a = (torch.randn(M, K, generator=g) * 2.0).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    aq, asc = ops.quant_rows_fp8(a)
    wq, wsc = ops.quant_rows_fp8(w)
    ad = aq.view(torch.float8_e4m3fn).float() * asc[:, None]
    wd = wq.view(torch.float8_e4m3fn).float() * wsc[:, None]
    exact = ad @ wd.t()
    b = torch.randn(N, generator=g).bfloat16().cuda()
    r = torch.randn(M, N, generator=g).bfloat16().cuda()
    gate = torch.randn(N, generator=g).cuda()
    rel = lambda x, y: ((x.float() - y).norm() / y.norm()).item()   # noqa: E731
    y = ops.linear_fp8(aq, asc, wq, wsc, b)
    assert rel(y, exact + b.float()) < 4e-3
    assert (y.float() - (exact + b.float())).abs().max().item() < 0.02 * exact.abs().max().item() + 0.05     # no stray tile / row / column
    assert rel(ops.linear_fp8(aq, asc, wq, wsc, b, epilogue='gelu'), torch.nn.functional.gelu(exact + b.float(), approximate='tanh')) < 5e-3
    assert rel(ops.linear_fp8(aq, asc, wq, wsc, b, epilogue='gate_res', gate=gate, residual=r), r.float() + gate * (exact + b.float())) < 4e-3
    out = torch.full((M + 1, N + 8), 7.0, dtype=torch.bfloat16, device='cuda')        # strided output view: nothing written outside it
    ops.linear_fp8(aq, asc, wq, wsc, b, out=out[:M, :N])
    assert rel(out[:M, :N], exact + b.float()) < 4e-3 and (out[M] == 7).all() and (out[:, N:] == 7).all()


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,K', [(4645, 3640, 1024), (300, 520, 512), (4608, 3072, 3072), (77, 3072, 15360)])
def test_fp8_block_scaled_quant_and_linear(M, N, K):
    """MX-style activations: one E8M0 byte per row and 128 columns (the fp8 kernel's K-tile), applied by the matrix instruction.  The
    quantiser against its definition (torch), the GEMM against the exact fp32 product of the dequantised operands -- every epilogue,
    ragged edges, a launch of a few tiles -- and the block scales must beat one scale per row on rows with an outlier."""
    from arcflow_amd import ops
    g = torch.Generator().manual_seed(7)
    a = (torch.randn(M, K, generator=g) * 2.0)
    a[::7, 5] = 2000.0                                             # outliers: one per affected row, in block 0 (their weight column is zero)
    a[:, 128:256] *= 1e-2                                          # a block of small values ...
    a = a.bfloat16().cuda()
    w = torch.randn(N, K, generator=g) * 0.05
    w[:, 128:256] *= 1e2                                           # ... that matter as much as the others
    w[:, 5] = 0
    w = w.bfloat16().cuda()
    aq, amx = ops.quant_rows_mx8(a)
    wq, wsc = ops.quant_rows_fp8(w)
    blocks = a.float().view(M, K // 128, 128)
    amax = blocks.abs().amax(-1)
    e = torch.ceil(torch.log2(amax.clamp_min(1e-30) / 448.0)).clamp(-126, 127)
    assert torch.equal(amx.cpu().to(torch.int32) - 127, e.cpu().to(torch.int32))
    scale = torch.exp2(e)[..., None]
    ref_codes = (blocks / scale).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).view(M, K)
    assert torch.equal(aq, ref_codes)                              # power-of-two scaling is exact: the codes are the RNE casts, bit for bit
    ad = (aq.view(torch.float8_e4m3fn).float().view(M, K // 128, 128) * scale).view(M, K)
    wd = wq.view(torch.float8_e4m3fn).float() * wsc[:, None]
    exact = ad @ wd.t()
    b = torch.randn(N, generator=g).bfloat16().cuda()
    r = torch.randn(M, N, generator=g).bfloat16().cuda()
    gate = torch.randn(N, generator=g).cuda()
    rel = lambda x, y: ((x.float() - y).norm() / y.norm()).item()   # noqa: E731
    y = ops.linear_fp8_mx(aq, amx, wq, wsc, b)
    assert rel(y, exact + b.float()) < 4e-3
    assert (y.float() - (exact + b.float())).abs().max().item() < 0.02 * exact.abs().max().item() + 0.05
    assert rel(ops.linear_fp8_mx(aq, amx, wq, wsc, b, epilogue='gelu'), torch.nn.functional.gelu(exact + b.float(), approximate='tanh')) < 5e-3
    assert rel(ops.linear_fp8_mx(aq, amx, wq, wsc, b, epilogue='gate_res', gate=gate, residual=r), r.float() + gate * (exact + b.float())) < 4e-3
    # against the unquantised product: with one scale per row the outlier pushes the small block of its row into e4m3's subnormals;
    # a block scale never sees another block's outlier
    true = a.float() @ w.float().t()
    q8, s8 = ops.quant_rows_fp8(a)
    per_row = ops.linear_fp8(q8, s8, wq, wsc)
    err_mx, err_row = rel(ops.linear_fp8_mx(aq, amx, wq, wsc)[::7], true[::7]), rel(per_row[::7], true[::7])
    assert err_mx < 6e-2 and err_mx < err_row, (err_mx, err_row)       # (one small block of K / 128: 0.7x at K = 1024, a few % at K = 15360)


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,K,col0,gelu', [(4645, 2048, 1024, 0, True), (1000, 1792, 512, 768, True), (300, 640, 1024, 128, False)])
def test_fp8_gemm_writes_block_scaled_operand(M, N, K, col0, gelu):
    """The producer epilogue: columns from col0 on leave the fp8 GEMM as e4m3 bytes + one scale byte per row and 128 columns, exactly what
    ``quant_rows_mx8`` makes of the same fp32 values (scale bytes equal; codes equal except where the bf16 detour of the two-pass path rounds
    differently), the columns before it as bf16.  Then chained: the written operand feeds the block-scaled GEMM."""
    from arcflow_amd import ops
    g = torch.Generator().manual_seed(11)
    a = (torch.randn(M, K, generator=g) * 2.0).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    b = torch.randn(N, generator=g).bfloat16().cuda()
    aq, amx = ops.quant_rows_mx8(a)
    wq, wsc = ops.quant_rows_fp8(w)
    scale_a = torch.exp2(amx.float() - 127)[..., None]
    ad = (aq.view(torch.float8_e4m3fn).float().view(M, K // 128, 128) * scale_a).view(M, K)
    wd = wq.view(torch.float8_e4m3fn).float() * wsc[:, None]
    y = ad @ wd.t() + b.float()
    if gelu:
        y[:, col0:] = torch.nn.functional.gelu(y[:, col0:], approximate='tanh')
    head, q, mx = ops.linear_fp8_to_mx8(aq, amx, wq, wsc, b, gelu=gelu, c8_col0=col0)
    if col0:
        assert ((head.float() - y[:, :col0]).norm() / y[:, :col0].norm()).item() < 4e-3
    n8 = N - col0
    blocks = y[:, col0:].reshape(M, n8 // 128, 128)
    e = torch.ceil(torch.log2(blocks.abs().amax(-1).clamp_min(1e-30) / 448.0)).clamp(-126, 127)
    got_e = mx.to(torch.int32) - 127
    assert (got_e != e.to(torch.int32)).float().mean().item() < 2e-3            # (a block maximum within an fp32 rounding of a power of two)
    deq = (q.view(torch.float8_e4m3fn).float().view(M, n8 // 128, 128) * torch.exp2(got_e.float())[..., None]).view(M, n8)
    assert ((deq - y[:, col0:]).norm() / y[:, col0:].norm()).item() < 4.5e-2       # e4m3: 3 mantissa bits
    assert (deq - y[:, col0:]).abs().max().item() <= 0.0625 * y[:, col0:].abs().max().item() + 1e-3
    # chained: the next GEMM consumes what this epilogue wrote
    if n8 % 512 == 0:
        w2 = (torch.randn(384, n8, generator=g) * 0.05).bfloat16().cuda()
        w2q, w2s = ops.quant_rows_fp8(w2)
        z = ops.linear_fp8_mx(q, mx, w2q, w2s)
        zref = deq @ (w2q.view(torch.float8_e4m3fn).float() * w2s[:, None]).t()
        assert ((z.float() - zref).norm() / zref.norm()).item() < 4e-3


@pytest.mark.gpu
@pytest.mark.parametrize('B,S,H', [(1, 4608, 3), (2, 333, 2)])
def test_attention_writes_block_scaled_operand(B, S, H):
    """The attention epilogue as a producer of the fp8 format: e4m3 bytes + one E8M0 byte per token and head, against the quantiser applied to the
    kernel's own bf16 output (scale bytes equal except where the bf16 rounding of the block maximum crosses a power of two; values equal to e4m3
    resolution) -- ragged S included."""
    from arcflow_amd import ops
    g = torch.Generator().manual_seed(13)
    q, k, v = ((torch.randn(B, S, H, 128, generator=g) * s).bfloat16().cuda() for s in (1.0, 1.0, 2.0))
    ops.set_attn_impl(3)                 # the fp8 epilogue runs on the plain grid: compare with the bf16 output of the same schedule (a merged row is rounded twice)
    o = ops.attention(q, k, v).reshape(B * S, H * 128)
    ops.set_attn_impl(0)
    o8, mx = ops.attention_to_mx8(q, k, v)
    rq, rmx = ops.quant_rows_mx8(o)
    assert (mx != rmx).float().mean().item() < 0.02
    deq = (o8.view(torch.float8_e4m3fn).float().view(B * S, H, 128) * torch.exp2(mx.float() - 127)[..., None]).view(B * S, H * 128)
    assert ((deq - o.float()).norm() / o.float().norm()).item() < 4.5e-2
    assert (deq - o.float()).abs().max().item() <= 0.0625 * o.float().abs().max().item() + 1e-3
    same = mx == rmx
    assert (o8.view(B * S, H, 128)[same] != rq.view(B * S, H, 128)[same]).float().mean().item() < 0.08     # the bf16 detour moves a value across an e4m3 rounding boundary now and then


# ------------------------------------------------------------------------------------------ stream-K tail of the GEMM
@pytest.mark.parametrize('M,N,K', [(4608, 3072, 3072), (4608, 3072, 15360), (4608, 9216, 3072), (4608, 12288, 3072),
                                   (2048, 1024, 512), (4608, 21504, 3072)])
def test_linear_stream_k_tail(ops, M, N, K):
    """The FLUX block GEMM shapes whose last round is under-filled (216 / 648 / 864 tiles on 256 CUs) with the stream-K tail
    on: against fp32 math on the device AND against the plain launch (same products, only the order of the fp32 partial sums
    differs).  Run twice on one workspace: the hand-off flags must be re-armed by the kernel."""
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device='cuda').bfloat16()
    w = (torch.randn(N, K, generator=g, device='cuda') * 0.03).bfloat16()
    b = torch.randn(N, generator=g, device='cuda').bfloat16()
    ws = ops.stream_k_workspace()
    plain = ops.linear(a, w, b, epilogue='gelu', gelu_col0=N // 2)
    ref = a.float() @ w.float().t() + b.float()
    ref[:, N // 2:] = torch.nn.functional.gelu(ref[:, N // 2:], approximate='tanh')
    from arcflow_amd import _lib
    # 32 tiles / 1512 tiles (last round 29 of 32 CUs per XCD busy): no under-filled last round worth splitting -> plain launch
    split_expected = (M, N, K) not in ((2048, 1024, 512), (4608, 21504, 3072))
    for rep in range(2):
        out = ops.linear(a, w, b, epilogue='gelu', gelu_col0=N // 2, sk_ws=ws)
        torch.cuda.synchronize()
        # the split path must really have run (the launcher used to ignore sk_force and these tests compared the plain
        # kernel with itself); afx_linear_sk_last_split = CUs per XCD the tail was split over, 0 = plain launch
        assert (_lib.load().afx_linear_sk_last_split() > 0) == split_expected
        assert rel_l2(out, ref) < 4e-3, rep
        assert rel_l2(out, plain) < 1e-3, rep
        assert (out.float() - plain.float()).abs().max().item() < 0.05 * plain.float().abs().max().item()
    flags = ws[:4096].view(torch.int32)
    assert int(flags.abs().sum()) == 0, 'hand-off flags not re-armed / timeout word set'


def test_linear_stream_k_gate_residual_inplace(ops):
    """mlp2 / proj_out form: C = res + gate * (A W^T + b) written over the residual, long K (the owner adds 1-2 partial slabs)."""
    g = torch.Generator(device='cuda').manual_seed(77)
    M, N, K = 4608, 3072, 12288
    a = torch.randn(M, K, generator=g, device='cuda').bfloat16()
    w = (torch.randn(N, K, generator=g, device='cuda') * 0.02).bfloat16()
    b = torch.randn(N, generator=g, device='cuda').bfloat16()
    gate = torch.randn(1, N, generator=g, device='cuda')
    x = torch.randn(M, N, generator=g, device='cuda').bfloat16()
    ref = x.float() + gate * (a.float() @ w.float().t() + b.float())
    ws = ops.stream_k_workspace()
    out = ops.linear(a, w, b, epilogue='gate_res', gate=gate, residual=x, rows_per_batch=M, out=x, sk_ws=ws)
    from arcflow_amd import _lib
    assert _lib.load().afx_linear_sk_last_split() > 0
    assert rel_l2(out, ref) < 4e-3



@pytest.mark.parametrize('out_f,in_f,r', [(384, 256, 32), (3072, 3072, 256), (200, 136, 8)])
def test_lora_merge_runs_on_the_library_gemm(out_f, in_f, r):
    """Load-time LoRA fold W + B A (arcflow_amd/weights.py, reference: arcflux.py:295-301 keeps the side GEMMs) goes through
    afx_linear_bf16_f32out on the device -- compared with the fp32 host product of the same bf16 adapters."""
    from arcflow_amd.weights import merge_lora
    g = torch.Generator().manual_seed(out_f + r)
    w = (torch.randn(out_f, in_f, generator=g) * 0.05).bfloat16()
    a = (torch.randn(r, in_f, generator=g) * 0.1).bfloat16()
    b = (torch.randn(out_f, r, generator=g) * 0.1).bfloat16()
    ref = (w.float() + b.float() @ a.float()).bfloat16()
    got = merge_lora({'m.weight': w.cuda()}, {'m.lora_A.weight': a.cuda(), 'm.lora_B.weight': b.cuda()})['m.weight']
    assert got.is_cuda and got.dtype == torch.bfloat16
    d = (got.cpu().float() - ref.float()).abs()
    # fp32 accumulation order differs: at most one bf16 ulp on a few entries
    assert d.max() <= 2.0 ** -7 * ref.float().abs().max() and (d > 0).float().mean() < 0.01


@pytest.mark.parametrize('rows,rpb', [(1025, 1025), (2050, 1025), (2049, 683), (1024, 512)])
def test_norm_modulate_two_rows_per_wave(ops, rows, rpb):
    """D = 3072, rows >= 1024: norm_modulate_rows_kernel<6, 2> (two rows per wave, modulation vectors kept in registers and reloaded when the
    sample changes between a wave's rows) -- odd row counts, a sample boundary inside a wave's pair, strided input."""
    D = 3072
    g = torch.Generator().manual_seed(rows + rpb)
    B = (rows + rpb - 1) // rpb
    xb = bf(torch.randn(rows, D + 64, generator=g) * 1.5 - 0.2).to(dev())
    x = xb[:, :D]                                              # row-strided view
    sc = torch.randn(B, D, generator=g).to(dev())
    sh = torch.randn(B, D, generator=g).to(dev())
    out = ops.norm_modulate(x, sc, sh, rows_per_batch=rpb)
    idx = torch.arange(rows, device=dev()) // rpb
    ref = torch.nn.functional.layer_norm(x.float(), (D,), eps=1e-6) * (1 + sc[idx]) + sh[idx]
    assert rel_l2(out, ref) < 4e-3
    assert (out.float() - ref).abs().max().item() < 0.08
