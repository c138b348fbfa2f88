"""GPU parity of the whole denoiser forward (HIP engine through the C ABI) against the fp32 CPU oracle
on identical bf16-rounded weights, reduced sizes (the oracle needs seconds)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2.5e-2     # rel-L2 of bf16 trunk outputs vs fp32 oracle (stated tolerance)


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp(min=1e-12)).item()


def _inputs(B, hp, wp, T, joint, pooled_dim, seed=1):
    g = torch.Generator().manual_seed(seed)
    hid = torch.randn(B, hp * wp, 64, generator=g).bfloat16()
    ctx = torch.randn(B, T, joint, generator=g).bfloat16()
    pooled = torch.randn(B, pooled_dim, generator=g).bfloat16() if pooled_dim else None
    return hid, ctx, pooled


@pytest.mark.parametrize('B,hp,wp,T,nd,ns', [(1, 8, 8, 16, 2, 2), (2, 6, 10, 7, 1, 3), (1, 16, 16, 77, 1, 0),
                                             (6, 4, 4, 5, 1, 1),       # batch 6: two micro-batches (4 + 2) inside the library
                                             (2, 6, 8, 16, 1, 2)])     # T % 16 == 0 and S = 64: V^T comes straight out of the projection
def test_flux_forward_vs_oracle(B, hp, wp, T, nd, ns):
    from arcflow_amd import MMDiTEngine
    from oracle import dit_ref as D
    cfg = D.FluxCfg(num_layers=nd, num_single_layers=ns, heads=2, joint_dim=128, pooled_dim=64)
    w = D.make_flux_weights(cfg, seed=3)
    hid, ctx, pooled = _inputs(B, hp, wp, T, 128, 64)
    t = torch.tensor([1.0, 0.7619, 0.5, 0.3, 0.9, 0.1][:B])
    gd = torch.full((B,), 3.5)
    rm, rlw, rlg = D.flux_forward(w, cfg, hid.float(), ctx.float(), pooled.float(), t, gd, hp, wp)
    eng = MMDiTEngine('flux', nd, ns, heads=2, joint_dim=128, pooled_dim=64)
    eng.load_state_dict(w)
    out = eng(hid.cuda(), t.cuda(), ctx.cuda(), pooled.cuda(), gd.cuda(), hp, wp)
    torch.cuda.synchronize()
    assert out.means.shape == rm.shape and out.logweights.shape == rlw.shape and out.loggammas.shape == rlg.shape
    assert rel_l2(out.means.float(), rm) < TOL
    assert rel_l2(out.loggammas.float(), rlg) < TOL
    assert (out.logweights.float().cpu() - rlw).abs().max().item() < 0.08
    # log_softmax over K is normalised
    assert torch.allclose(out.logweights.float().exp().sum(dim=2).cpu(), torch.ones(B, hp * wp, 4), atol=2e-2)


def test_flux_forward_8phase_gemm_and_separate_qk_prep_launch():
    """The default forward runs the one-wave-per-SIMD GEMM with q / k RMSNorm + RoPE fused into the k|v|q projections' epilogue
    (GemmProblem::qk_D).  With the 8-phase kernel selected the engine falls back to the separate kv_prep launch: same parity bar,
    and the two paths agree with each other (the fused one normalises the fp32 accumulators, the separate one bf16-rounded q / k)."""
    from arcflow_amd import MMDiTEngine, ops
    from oracle import dit_ref as D
    cfg = D.FluxCfg(num_layers=1, num_single_layers=2, heads=2, joint_dim=128, pooled_dim=64)
    w = D.make_flux_weights(cfg, seed=5)
    hid, ctx, pooled = _inputs(2, 8, 8, 9, 128, 64, seed=4)
    t, gd = torch.tensor([0.9, 0.4]), torch.full((2,), 3.5)
    rm, rlw, rlg = D.flux_forward(w, cfg, hid.float(), ctx.float(), pooled.float(), t, gd, 8, 8)
    eng = MMDiTEngine('flux', 1, 2, heads=2, joint_dim=128, pooled_dim=64)
    eng.load_state_dict(w)
    fused = eng(hid.cuda(), t.cuda(), ctx.cuda(), pooled.cuda(), gd.cuda(), 8, 8)
    try:
        ops.set_gemm_mode(2, 0)
        plain = eng(hid.cuda(), t.cuda(), ctx.cuda(), pooled.cuda(), gd.cuda(), 8, 8)
    finally:
        ops.set_gemm_mode(3, 0)
    torch.cuda.synchronize()
    for out in (fused, plain):
        assert rel_l2(out.means.float(), rm) < TOL and rel_l2(out.loggammas.float(), rlg) < TOL
    assert rel_l2(fused.means.float(), plain.means.float()) < 1.5e-2
    assert not torch.equal(fused.means, plain.means)          # (they ARE different code paths)


def test_prepared_steps_are_bit_identical_to_the_plain_forward():
    """afx_mmdit_prepare_steps: the modulation vectors of several steps from ONE pass over the stacked modulation matrix.  A forward
    that takes a prepared step must equal the plain forward at that timestep bit for bit (same kernels, same operands), for
    every step and sample, and fall back to the plain path when nothing matching is prepared."""
    from arcflow_amd import MMDiTEngine
    from oracle import dit_ref as D
    cfg = D.FluxCfg(num_layers=1, num_single_layers=1, heads=2, joint_dim=128, pooled_dim=64)
    w = D.make_flux_weights(cfg, seed=6)
    B = 2
    hid, ctx, pooled = _inputs(B, 8, 8, 9, 128, 64, seed=7)
    gd = torch.full((B,), 3.5)
    sig = [1.0, 0.7619, 0.3]
    eng = MMDiTEngine('flux', 1, 1, heads=2, joint_dim=128, pooled_dim=64)
    eng.load_state_dict(w)
    plain = [eng(hid.cuda(), torch.full((B,), s).cuda(), ctx.cuda(), pooled.cuda(), gd.cuda(), 8, 8) for s in sig]
    assert eng.prepare_steps(sig, pooled, gd, B, 64, 9)
    for k in (2, 0, 1):                                  # any order
        out = eng(hid.cuda(), torch.zeros(B).cuda(), ctx.cuda(), pooled.cuda(), gd.cuda(), 8, 8, prepared_step=k)    # (t argument unused)
        for name in ('means', 'logweights', 'loggammas'):
            assert torch.equal(getattr(out, name), getattr(plain[k], name)), (k, name)
    # one-shot: the next plain call recomputes from its own t
    again = eng(hid.cuda(), torch.full((B,), sig[1]).cuda(), ctx.cuda(), pooled.cuda(), gd.cuda(), 8, 8)
    assert torch.equal(again.means, plain[1].means)
    # a batch that was not prepared falls back to the plain path (and is right)
    one = eng(hid[:1].cuda(), torch.full((1,), sig[0]).cuda(), ctx[:1].cuda(), pooled[:1].cuda(), gd[:1].cuda(), 8, 8, prepared_step=0)
    assert torch.equal(one.means, plain[0].means[:1])
    assert not eng.prepare_steps([0.5] * 5, pooled, gd, B, 64, 9)       # 5 steps x 2 samples > 8 rows: nothing prepared


def test_flux_teacher_head():
    from arcflow_amd import MMDiTEngine
    from oracle import dit_ref as D
    cfg = D.FluxCfg(num_layers=1, num_single_layers=1, heads=2, joint_dim=128, pooled_dim=64)
    w = D.make_flux_weights(cfg, seed=4, teacher_head=True)
    hid, ctx, pooled = _inputs(1, 8, 8, 12, 128, 64, seed=2)
    t, gd = torch.tensor([0.6]), torch.tensor([3.5])
    ref = D.flux_teacher_forward(w, cfg, hid.float(), ctx.float(), pooled.float(), t, gd, 8, 8)
    eng = MMDiTEngine('flux', 1, 1, heads=2, joint_dim=128, pooled_dim=64, teacher_head=True)
    eng.load_state_dict(w)
    out = eng(hid.cuda(), t.cuda(), ctx.cuda(), pooled.cuda(), gd.cuda(), 8, 8)
    assert out.shape == ref.shape
    assert rel_l2(out.float(), ref) < TOL


@pytest.mark.parametrize('B,hp,wp,T,nl', [(1, 8, 8, 11, 2), (2, 6, 8, 20, 1)])
def test_qwen_forward_vs_oracle(B, hp, wp, T, nl):
    from arcflow_amd import MMDiTEngine
    from oracle import dit_ref as D
    cfg = D.QwenCfg(num_layers=nl, heads=2, joint_dim=192)
    w = D.make_qwen_weights(cfg, seed=5)
    hid, ctx, _ = _inputs(B, hp, wp, T, 192, 0)
    t = torch.tensor([1.0, 0.5][:B])
    rm, rlw, rlg = D.qwen_forward(w, cfg, hid.float(), ctx.float(), t, hp, wp)
    eng = MMDiTEngine('qwen', nl, 0, heads=2, joint_dim=192)
    eng.load_state_dict(w)
    out = eng(hid.cuda(), t.cuda(), ctx.cuda(), None, None, hp, wp)
    assert rel_l2(out.means.float(), rm) < TOL
    assert rel_l2(out.loggammas.float(), rlg) < TOL
    assert (out.logweights.float().cpu() - rlw).abs().max().item() < 0.08


def test_engine_errors_are_loud():
    from arcflow_amd import MMDiTEngine, _lib
    eng = MMDiTEngine('flux', 1, 0, heads=2, joint_dim=128, pooled_dim=64)
    with pytest.raises(_lib.ArcflowHipError):      # nothing bound yet
        eng(torch.zeros(1, 4, 64).cuda(), torch.ones(1).cuda(), torch.zeros(1, 3, 128).cuda(),
            torch.zeros(1, 64).cuda(), torch.ones(1).cuda(), 2, 2)


@pytest.mark.parametrize('family', ['flux', 'qwen'])
def test_fp8_linear_mode_tracks_bf16(family):
    """BASELINE.json configs[4] "fp8 MFMA fwd": the same engine with every block linear on the fp8 MFMA (row-wise e4m3
    scales) stays within fp8 quantisation error of the bf16 forward; the teacher head likewise."""
    from arcflow_amd import MMDiTEngine
    from oracle import dit_ref as D
    g = torch.Generator().manual_seed(21)
    hp = wp = 8
    T = 24
    if family == 'flux':
        cfg = D.FluxCfg(num_layers=2, num_single_layers=2, heads=2, joint_dim=128, pooled_dim=64)
        w = D.make_flux_weights(cfg, seed=5)
        kw = dict(num_double=2, num_single=2, heads=2, joint_dim=128, pooled_dim=64)
        pooled, gd = (torch.randn(1, 64, generator=g) * 0.5).bfloat16().cuda(), torch.full((1,), 3.5).cuda()
    else:
        cfg = D.QwenCfg(num_layers=3, heads=2, joint_dim=192)
        w = D.make_qwen_weights(cfg, seed=5)
        kw = dict(num_double=3, heads=2, joint_dim=192)
        pooled = gd = None
    x = torch.randn(1, hp * wp, 64, generator=g).bfloat16().cuda()
    ctx = (torch.randn(1, T, kw['joint_dim'], generator=g) * 0.5).bfloat16().cuda()
    t = torch.tensor([0.6]).cuda()
    outs = []
    for fp8 in (False, True):
        eng = MMDiTEngine(family, kw['num_double'], kw.get('num_single', 0), heads=2, joint_dim=kw['joint_dim'], pooled_dim=kw.get('pooled_dim', 768))
        eng.load_state_dict(w)
        if fp8:
            eng.enable_fp8()
        outs.append(eng(x, t, ctx, pooled, gd, hp, wp))
    for k in ('means', 'logweights', 'loggammas'):
        a, b = outs[0][k].float(), outs[1][k].float()
        rel = ((a - b).norm() / a.norm()).item()
        assert 0 < rel < 8e-2, (k, rel)          # different (fp8) but close


@pytest.mark.gpu
@pytest.mark.parametrize('family', ['flux', 'qwen'])
def test_fp8_block_scaled_mode_tracks_bf16(family, monkeypatch):
    """Width 512 (4 heads) switches the fp8 engine to block-scaled activations (one E8M0 byte per row and 128 columns; the mlp hidden and the mlp part
    of the single blocks' [O | mlp] operand leave the producing GEMM's epilogue quantised, no pass of their own).  Against the bf16 engine and
    against the same engine with one scale per row and a quantisation pass per GEMM (AFX_FP8_MX=0): both within e4m3 error, and not identical."""
    from arcflow_amd import MMDiTEngine
    from oracle import dit_ref as D
    g = torch.Generator().manual_seed(23)
    hp = wp = 16
    T = 40
    if family == 'flux':
        cfg = D.FluxCfg(num_layers=2, num_single_layers=2, heads=4, joint_dim=128, pooled_dim=64)
        w = D.make_flux_weights(cfg, seed=6)
        kw = dict(num_double=2, num_single=2, joint_dim=128, pooled_dim=64)
        pooled, gd = (torch.randn(2, 64, generator=g) * 0.5).bfloat16().cuda(), torch.full((2,), 3.5).cuda()
    else:
        cfg = D.QwenCfg(num_layers=3, heads=4, joint_dim=192)
        w = D.make_qwen_weights(cfg, seed=6)
        kw = dict(num_double=3, joint_dim=192)
        pooled = gd = None
    x = torch.randn(2, hp * wp, 64, generator=g).bfloat16().cuda()
    ctx = (torch.randn(2, T, kw['joint_dim'], generator=g) * 0.5).bfloat16().cuda()
    t = torch.tensor([0.6, 0.3]).cuda()
    outs = {}
    for mode in ('bf16', 'mx', 'row'):
        monkeypatch.setenv('AFX_FP8_MX', '0' if mode == 'row' else '1')
        eng = MMDiTEngine(family, kw['num_double'], kw.get('num_single', 0), heads=4, joint_dim=kw['joint_dim'], pooled_dim=kw.get('pooled_dim', 768))
        eng.load_state_dict(w)
        if mode != 'bf16':
            eng.enable_fp8()
        outs[mode] = {k: v.float().clone() for k, v in eng(x, t, ctx, pooled, gd, hp, wp).items()}
    for k in ('means', 'logweights', 'loggammas'):
        ref = outs['bf16'][k]
        rel_mx = ((outs['mx'][k] - ref).norm() / ref.norm()).item()
        rel_row = ((outs['row'][k] - ref).norm() / ref.norm()).item()
        assert 0 < rel_mx < 8e-2 and 0 < rel_row < 8e-2, (k, rel_mx, rel_row)
        assert rel_mx < 1.3 * rel_row + 1e-3, (k, rel_mx, rel_row)
        assert not torch.equal(outs['mx'][k], outs['row'][k])          # the two formats really are different paths


@pytest.mark.gpu
def test_profile_events_every_launch_and_sampled():
    """afx_profile_enable(ctx, N): an event pair on every GEMM / attention launch (N = 1) or on one launch in N (bench.py's default 8: the pairs cost
    ~4 us each).  The sampled sums must cover 1 / N of the launches and give the same FLOP / time ratio within the launch-to-launch spread."""
    from arcflow_amd import MMDiTEngine
    from oracle import dit_ref as D
    cfg = D.FluxCfg(num_layers=2, num_single_layers=3, heads=2, joint_dim=128, pooled_dim=64)
    w = D.make_flux_weights(cfg, seed=0)
    eng = MMDiTEngine('flux', 2, 3, heads=2, joint_dim=128, pooled_dim=64)
    eng.load_state_dict(w)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 256, 64, generator=g).bfloat16().cuda()
    ctx = torch.randn(1, 64, 128, generator=g).bfloat16().cuda()
    pooled = torch.randn(1, 64, generator=g).bfloat16().cuda()
    t, gd = torch.tensor([0.5]).cuda(), torch.tensor([3.5]).cuda()

    def run(n, stride):
        eng.profile(stride)
        for _ in range(n):
            eng(x, t, ctx, pooled, gd, 16, 16)
        torch.cuda.synchronize()
        r = eng.profile_read(0), eng.profile_read(1)
        eng.profile(False)
        return r
    (ms_g, n_g, fl_g), (ms_a, n_a, fl_a) = run(6, 1)
    assert n_a == 6 * 5 and n_g > n_a and ms_g > 0 and ms_a > 0 and fl_g > 0        # one attention launch per block and forward
    per_fwd = (n_g + n_a) // 6
    (ms_g3, n_g3, fl_g3), (ms_a3, n_a3, fl_a3) = run(6, 3)
    assert n_g3 + n_a3 == (6 * per_fwd + 2) // 3                                     # launches 0, 3, 6, ... of the counter over both classes
    assert 0.2 < (n_g3 / max(n_g, 1)) < 0.5 and abs(fl_g3 / n_g3 - fl_g / n_g) < 0.5 * fl_g / n_g
    out = eng(x, t, ctx, pooled, gd, 16, 16)                                         # profiling off again: nothing recorded
    torch.cuda.synchronize()
    assert eng.profile_read(0)[1] == 0 and torch.isfinite(out.means.float()).all()
