"""GPU parity of the distillation-step kernels against torch autograd on the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from arcflow_amd import ops as _ops
    return _ops


def close(a, b, rtol=2e-5, atol=2e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f'max abs err {err}, ref scale {b.abs().max().item()}'


def _mix(seed=0, B=2, N=24, K=16, ch=64, pp=4):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, N, ch, generator=g)
    m = torch.randn(B, N, K, ch, generator=g)
    lw = torch.log_softmax(torch.randn(B, N, K, pp, generator=g) * 1.5, dim=2)
    lg = torch.randn(B, N, K - 1, pp, generator=g)
    lg[0, 0] = 0.0
    lg[0, 1] = 3e-5
    lg[1, 2] = -3e-5
    up = torch.randn(B, N, ch, generator=g)
    return x, m, lw, lg, up


def test_arcflow_backward_step_and_velocity(ops):
    from oracle import arcflow_ref as R
    x, m, lw, lg, up = _mix()
    B = x.shape[0]
    s_src = torch.tensor([1.0, 0.8])
    s_a = torch.tensor([0.9, 0.55])
    s_b = torch.tensor([0.4, 0.1])
    gs = torch.tensor([0.7, -1.3])
    mr, lwr, lgr = (t.clone().requires_grad_(True) for t in (m, lw, lg))
    tot = 0
    for b in range(B):
        xe = R.momentum_step_packed(x[b:b + 1], mr[b:b + 1], lwr[b:b + 1], lgr[b:b + 1], float(s_src[b]), float(s_a[b]), float(s_b[b]))
        disp = x[b:b + 1] - xe
        tot = tot + (disp * up[b:b + 1]).sum() * gs[b]
    tot.backward()
    dm, dw, dg = ops.arcflow_backward(up.cuda(), m.cuda(), lw.cuda(), lg.cuda(), s_src.cuda(), s_a.cuda(), s_b.cuda(), gscale=gs.cuda())
    close(dm, mr.grad)
    close(dw, lwr.grad, atol=2e-4)
    close(dg, lgr.grad, atol=2e-4)
    # velocity form + accumulation into the same buffers
    mr2, lwr2, lgr2 = (t.clone().requires_grad_(True) for t in (m, lw, lg))
    hp, wp = 4, 6
    ml, lwl, lgl = R.unpack_mixture(mr2, lwr2, lgr2, hp, wp)
    u = R.pack_latents(R.policy_velocity(ml, lwl, lgl, s_src, s_a))
    (u * up).sum().backward()
    ops.arcflow_backward(up.cuda(), m.cuda(), lw.cuda(), lg.cuda(), s_src.cuda(), s_a.cuda(), s_a.cuda(), velocity=True, grads=(dm, dw, dg))
    close(dm, mr.grad + mr2.grad)
    close(dw, lwr.grad + lwr2.grad, atol=2e-4)
    close(dg, lgr.grad + lgr2.grad, atol=2e-4)


def test_step_dropout_euler_cfg_mse(ops):
    from oracle import arcflow_ref as R
    x, m, lw, lg, up = _mix(1)
    B, N, K = x.shape[0], x.shape[1], m.shape[2]
    drop = torch.zeros(B, K, dtype=torch.bool)
    drop[0, 3] = drop[0, 7] = drop[1, 0] = True
    s_src, s_a, s_b = torch.tensor([1.0, 0.7]), torch.tensor([0.95, 0.7]), torch.tensor([0.5, 0.2])
    out = ops.arcflow_step_dropout(x.cuda(), m.cuda(), lw.cuda(), lg.cuda(), s_src.cuda(), s_a.cuda(), s_b.cuda(), drop.cuda())
    lwm = lw.masked_fill(drop[:, None, :, None], float('-inf'))
    for b in range(B):
        ref = R.momentum_step_packed(x[b:b + 1], m[b:b + 1], lwm[b:b + 1], lg[b:b + 1], float(s_src[b]), float(s_a[b]), float(s_b[b]))
        close(out[b:b + 1], ref)
    close(ops.euler_roll(x.cuda(), up.cuda(), s_a.cuda(), s_b.cuda()), x + up * (s_b - s_a).reshape(B, 1, 1))
    close(ops.cfg_combine(x.cuda(), up.cuda(), 4.0), x + R.cfg_bias(x, up, 4.0))
    loss = torch.zeros(1, device='cuda')
    coef = 30.0 / x[0].numel() / B * 0.5
    g = ops.mse_loss(x.cuda(), up.cuda(), coef, loss)
    ref_loss = R.flow_mse_loss(x.reshape(B, -1), up.reshape(B, -1)) * 0.5        # x segment size 0.5
    close(loss[0], ref_loss, rtol=1e-5)
    close(g, coef * (x - up))


def test_head_grad_and_weight_grads(ops):
    g = torch.Generator().manual_seed(3)
    B, N, K, ch, lw_ch, D = 2, 96, 16, 64, 4, 256
    M = B * N
    xn = torch.randn(M, D, generator=g).bfloat16()
    w = (torch.randn(1152, D, generator=g) * 0.05).bfloat16()
    w[1148:] = 0
    wr = w.float().clone().requires_grad_(True)
    y = xn.float() @ wr.T
    means = y[:, :1024].reshape(B, N, K, ch)
    logw = y[:, 1024:1088].reshape(B, N, K, lw_ch).log_softmax(dim=2)
    logg = y[:, 1088:1148].reshape(B, N, K - 1, lw_ch)
    dm = torch.randn(B, N, K, ch, generator=g)
    dw = torch.randn(B, N, K, lw_ch, generator=g)
    dg = torch.randn(B, N, K - 1, lw_ch, generator=g)
    ((means * dm).sum() + (logw * dw).sum() + (logg * dg).sum()).backward()
    dy = ops.head_grad(dm.cuda(), dw.cuda(), dg.cuda(), logw.detach().bfloat16().cuda(), 1152)
    # reference dY from autograd: recompute via y.grad is not retained, so rebuild analytically
    yl = y.detach().clone().requires_grad_(True)
    ml = yl[:, :1024].reshape(B, N, K, ch)
    ll = yl[:, 1024:1088].reshape(B, N, K, lw_ch).log_softmax(dim=2)
    gl = yl[:, 1088:1148].reshape(B, N, K - 1, lw_ch)
    ((ml * dm).sum() + (ll * dw).sum() + (gl * dg).sum()).backward()
    close(dy.float()[:, :1148], yl.grad[:, :1148], rtol=1e-2, atol=2e-2)      # bf16 rows
    assert dy[:, 1148:].abs().max().item() == 0
    # dW = dY^T X through the transposed-operand GEMM with fp32 output
    dyt, xt = ops.transpose(dy), ops.transpose(xn.cuda())
    assert torch.equal(dyt.cpu(), dy.cpu().T) and torch.equal(xt.cpu(), xn.T)
    Mp = (M + 63) // 64 * 64
    assert Mp == M
    dW = ops.linear_f32out(dyt, xt)
    ref = dy.float().cpu().T @ xn.float()
    close(dW, ref, rtol=1e-4, atol=1e-3)
    dW2 = ops.linear_f32out(dyt, xt, out=dW.clone(), accumulate=True)
    close(dW2, 2 * ref, rtol=1e-4, atol=2e-3)
    rel = ((dW[:1148].cpu() - wr.grad[:1148]).norm() / wr.grad[:1148].norm()).item()
    assert rel < 5e-3, rel                                                       # vs autograd (dY rounded to bf16)
    db = ops.colsum(dy, torch.zeros(1152, device='cuda'))
    close(db, dy.float().cpu().sum(0), rtol=1e-5, atol=1e-3)


def test_normout_backward_and_outer(ops):
    g = torch.Generator().manual_seed(4)
    B, N, D = 2, 64, 512
    x = (torch.randn(B * N, D, generator=g) * 2 + 0.5).bfloat16()
    dxn = torch.randn(B * N, D, generator=g).bfloat16()
    sc = torch.randn(B, D, generator=g).requires_grad_(True)
    sh = torch.randn(B, D, generator=g).requires_grad_(True)
    xn = torch.nn.functional.layer_norm(x.float(), (D,), eps=1e-6).reshape(B, N, D) * (1 + sc[:, None]) + sh[:, None]
    (xn * dxn.float().reshape(B, N, D)).sum().backward()
    dmod = ops.normout_backward(x.cuda(), dxn.cuda(), torch.zeros(B, 2, D, device='cuda'), N)
    close(dmod[:, 0], sc.grad, rtol=1e-4, atol=1e-3)
    close(dmod[:, 1], sh.grad, rtol=1e-4, atol=1e-3)
    semb = torch.randn(B, 128, generator=g)
    dflat = dmod.reshape(B, 2 * D)
    dW = ops.outer_accum(dflat, semb.cuda(), torch.ones(2 * D, 128, device='cuda'))
    close(dW, 1 + dflat.cpu().T @ semb, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('B,N,D', [(1, 4608, 3072), (2, 516, 3072), (3, 100, 1024)])
def test_normout_backward_training_shapes(ops, B, N, D):
    """The two-stage version (per-wave partial sums, then a fold over the waves of a batch entry) at the training shapes: many waves per
    batch entry (split fold), a row count that forces fewer rows per wave (516 = 4 x 129), accumulation INTO a non-zero result."""
    g = torch.Generator().manual_seed(B * N)
    x = (torch.randn(B * N, D, generator=g) * 2 + 0.5).bfloat16()
    dxn = torch.randn(B * N, D, generator=g).bfloat16()
    xf = x.double().reshape(B, N, D)
    ln = (xf - xf.mean(-1, keepdim=True)) / torch.sqrt(xf.var(-1, unbiased=False, keepdim=True) + 1e-6)
    dsc = (dxn.double().reshape(B, N, D) * ln).sum(1)
    dsh = dxn.double().reshape(B, N, D).sum(1)
    start = torch.full((B, 2, D), 0.5, device='cuda')
    dmod = ops.normout_backward(x.cuda(), dxn.cuda(), start.clone(), N)
    close(dmod[:, 0] - 0.5, dsc.float(), rtol=2e-4, atol=2e-2)
    close(dmod[:, 1] - 0.5, dsh.float(), rtol=2e-4, atol=2e-2)


def test_adamw_ema_sumsq_cast(ops):
    g = torch.Generator().manual_seed(5)
    n = 100003
    p0 = torch.randn(n, generator=g)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01)
    p = p0.clone().cuda()
    m, v = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        pr.grad = gr.clone() * 0.5
        opt.step()
        ops.adamw_step(p, gr.cuda(), m, v, 1e-3, step, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01, grad_scale=0.5)
    close(p, pr.data, rtol=1e-5, atol=1e-6)
    ema = torch.randn(n, generator=g)
    e = ema.clone().cuda()
    ops.ema_lerp(e, p, 0.93)
    close(e, p.cpu() + (ema - p.cpu()) * 0.93)
    acc = ops.sumsq(p, torch.zeros(1, device='cuda'))
    close(acc[0], (p.cpu().double() ** 2).sum().float(), rtol=1e-4, atol=1e-2)
    close(ops.cast_bf16(p).float(), p.cpu().bfloat16().float(), rtol=0, atol=0)


@pytest.mark.parametrize('B,S,H', [(1, 64, 1), (2, 200, 2), (1, 333, 3), (1, 1024, 2), (2, 128, 1), (1, 97, 9), (1, 1500, 2), (1, 4608, 1)])
def test_attention_backward(ops, B, S, H):
    g = torch.Generator().manual_seed(S + 1)
    q, k, v, do = (torch.randn(B, S, H, 128, generator=g).bfloat16() for _ in range(4))
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qr.transpose(1, 2), kr.transpose(1, 2), vr.transpose(1, 2)).transpose(1, 2)
    (ref * do.float()).sum().backward()
    o, lse = ops.attention_fwd_lse(q.cuda(), k.cuda(), v.cuda())
    o_plain = ops.attention(q.cuda(), k.cuda(), v.cuda())
    assert torch.equal(o, o_plain)                                  # the LSE side output does not perturb O
    # lse is the log2-domain log-sum-exp of the scaled scores
    s = torch.einsum('bqhd,bkhd->bhqk', q.float(), k.float()) / 128 ** 0.5
    ref_lse = torch.logsumexp(s, dim=-1) / np.log(2.0)
    assert torch.allclose(lse[:, :, :S].cpu(), ref_lse, atol=2e-2)
    dq, dk, dv = ops.attention_bwd(q.cuda(), k.cuda(), v.cuda(), o.reshape(B, S, H, 128), do.cuda(), lse)

    def rel(a, b):
        return ((a.float().cpu() - b).norm() / b.norm()).item()
    assert rel(dv, vr.grad) < 1.5e-2, rel(dv, vr.grad)
    assert rel(dq, qr.grad) < 2e-2, rel(dq, qr.grad)
    assert rel(dk, kr.grad) < 2e-2, rel(dk, kr.grad)
    for t in (dq, dk, dv):
        assert torch.isfinite(t.float()).all()


@pytest.mark.parametrize('S,H', [(4608, 24), (4224, 24), (4133, 5)])
def test_attention_backward_generations_agree_at_the_production_shapes(ops, S, H):
    """FLUX (4096 + 512 tokens) and Qwen-Image (4096 + 128) shapes, 24 heads, and a ragged one: the generated streams (one launch / two launches / dK-dV only)
    against the round-4 kernels -- the same mathematics and the same bf16 rounding points (P and dS rounded to bf16 before the accumulating products), so the
    results differ by summation order only."""
    g = torch.Generator(device='cuda').manual_seed(S)
    q, k, v, do = (torch.randn(1, S, H, 128, generator=g, device='cuda').bfloat16() for _ in range(4))
    o, lse = ops.attention_fwd_lse(q, k, v)
    try:
        ops.set_attn_bwd_impl(2)
        ref = ops.attention_bwd(q, k, v, o.reshape(1, S, H, 128), do, lse)
        for impl in (3, 4, 1):
            ops.set_attn_bwd_impl(impl)
            got = ops.attention_bwd(q, k, v, o.reshape(1, S, H, 128), do, lse)
            for name, a, b in zip(('dq', 'dk', 'dv'), got, ref):
                assert torch.isfinite(a.float()).all()
                err = ((a.float() - b.float()).norm() / b.float().norm()).item()
                assert err < 6e-3, (impl, name, err)
    finally:
        ops.set_attn_bwd_impl(3)


def test_attention_backward_strided_views_and_determinism(ops):
    """The training trunk hands the backward row-strided views (q | k | v columns of one [rows, 3 H 128] stash, gradients into a second one); the generated
    kernels (afx_attn_bwd3.hip) read them through LDS-DMA with the caller's strides.  Same numbers as the contiguous call, bit for bit, and twice the same
    (no atomics anywhere in the backward)."""
    B, S, H = 2, 333, 2
    g = torch.Generator().manual_seed(11)
    q, k, v, do = (torch.randn(B, S, H, 128, generator=g).bfloat16().cuda() for _ in range(4))
    o, lse = ops.attention_fwd_lse(q, k, v)
    ref = ops.attention_bwd(q, k, v, o.reshape(B, S, H, 128), do, lse)
    D = H * 128
    stash = torch.zeros(B * S, 3 * D + 64, dtype=torch.bfloat16, device='cuda')
    stash[:, :D], stash[:, D:2 * D], stash[:, 2 * D:3 * D] = q.reshape(B * S, D), k.reshape(B * S, D), v.reshape(B * S, D)
    gbuf = torch.full((B * S, 3 * D), 7.0, dtype=torch.bfloat16, device='cuda')
    dov = torch.zeros(B * S, D + 8, dtype=torch.bfloat16, device='cuda')
    dov[:, :D] = do.reshape(B * S, D)
    for _ in range(2):
        ops.attention_bwd_2d(stash[:, :D], stash[:, D:2 * D], stash[:, 2 * D:3 * D], o.reshape(B * S, D), dov[:, :D], lse, gbuf[:, :D], gbuf[:, D:2 * D],
                             gbuf[:, 2 * D:], B, S, H)
        for i, r in enumerate(ref):
            assert torch.equal(gbuf[:, i * D:(i + 1) * D], r.reshape(B * S, D)), i


def test_elementwise_backward_kernels(ops):
    from oracle import dit_ref as D
    g = torch.Generator().manual_seed(9)
    B, S, Dm = 2, 37, 512
    x = (torch.randn(B * S, Dm, generator=g) * 1.5 + 0.2).bfloat16()
    dxn = torch.randn(B * S, Dm, generator=g).bfloat16()
    dres = torch.randn(B * S, Dm, generator=g).bfloat16()
    sc = torch.randn(B, Dm, generator=g)
    xr = x.float().clone().requires_grad_(True)
    xn = torch.nn.functional.layer_norm(xr, (Dm,), eps=1e-6).reshape(B, S, Dm) * (1 + sc[:, None])
    (xn * dxn.float().reshape(B, S, Dm)).sum().backward()
    out = ops.ln_modulate_backward(x.cuda(), dxn.cuda(), sc.cuda(), S, dres=dres.cuda())
    ref = xr.grad + dres.float()
    assert ((out.float().cpu() - ref).norm() / ref.norm()).item() < 6e-3
    # RMSNorm + RoPE out-of-place forward == in-place kernel, backward vs autograd
    hp, wp, T, H = 4, 5, 6, 2
    S2 = T + hp * wp
    q = torch.randn(B, S2, H * 128, generator=g).bfloat16()
    dy = torch.randn(B, S2, H * 128, generator=g).bfloat16()
    wt, wi = 1 + 0.1 * torch.randn(128, generator=g), 1 + 0.1 * torch.randn(128, generator=g)
    cos, sin = D.flux_rope_tables(hp, wp, T)
    y = ops.qk_norm_rope(q.cuda(), wt.cuda(), wi.cuda(), cos.cuda(), sin.cuda(), T)
    y_ip = ops.qk_norm_rope_(q.cuda().reshape(B, S2, H, 128).clone(), wt.cuda(), wi.cuda(), cos.cuda(), sin.cuda(), T)
    assert torch.equal(y.reshape(B, S2, H, 128), y_ip)
    qr = q.float().clone().requires_grad_(True)
    q4 = qr.reshape(B, S2, H, 128)
    yr = torch.cat([D.apply_rope(D.rms_norm(q4[:, :T], wt), cos[:T], sin[:T]),
                    D.apply_rope(D.rms_norm(q4[:, T:], wi), cos[T:], sin[T:])], dim=1)
    (yr * dy.float().reshape(B, S2, H, 128)).sum().backward()
    dq = ops.qk_norm_rope(q.cuda(), wt.cuda(), wi.cuda(), cos.cuda(), sin.cuda(), T, dy=dy.cuda())
    assert ((dq.float().cpu() - qr.grad).norm() / qr.grad.norm()).item() < 6e-3
    # GELU forward / backward, add_scale
    pre = (torch.randn(50, 256, generator=g) * 2).bfloat16()
    dh = torch.randn(50, 256, generator=g).bfloat16()
    pr = pre.float().clone().requires_grad_(True)
    hr = torch.nn.functional.gelu(pr, approximate='tanh')
    (hr * dh.float()).sum().backward()
    assert ((ops.gelu(pre.cuda()).float().cpu() - hr.detach()).abs().max().item()) < 2e-2
    dpre = ops.gelu(pre.cuda(), dh=dh.cuda())
    assert ((dpre.float().cpu() - pr.grad).norm() / pr.grad.norm()).item() < 6e-3
    gate = torch.randn(2, 256, generator=g)
    out = ops.add_scale(pre.cuda(), dh.cuda(), gate.cuda(), rows_per_batch=25)
    ref = (pre.float() + dh.float()) * gate.repeat_interleave(25, 0)
    assert ((out.float().cpu() - ref).norm() / ref.norm()).item() < 6e-3


# ------------------------------------------------------------------------------------------ TN product (LoRA weight gradients, afx_tn.hip)
@pytest.mark.parametrize('M,N1,N2', [(4608, 3072, 256), (4096, 256, 3072), (512, 12288, 256), (77, 256, 1024), (130, 64, 192), (4608, 256, 15360),
                                     (64, 128, 128), (1, 8, 8), (333, 200, 72)])
def test_linear_tn_f32out_matches_transposed_products(ops, M, N1, N2):
    """C (+)= X^T Y straight from token-major operands (ds_read_b64_tr_b16 fragments) against fp32 math and against the round-4 path (two explicit
    transposes + the NT kernel): the LoRA shapes (dB [out, 256] over 4608 / 512 tokens, dA [256, in]), ragged token counts (the last K-step's rows
    past M are zeroed in LDS), widths below / off the 128-column tile, strided views (a column range of a wider buffer) and the accumulate mode.
    The inputs are NOT symmetric: a swapped operand or a transposed output would show."""
    g = torch.Generator(device='cuda').manual_seed(M + 3 * N1 + 7 * N2)
    xw = torch.randn(M, N1 + 64, generator=g, device='cuda').bfloat16()
    yw = (torch.randn(M, N2 + 8, generator=g, device='cuda') * 0.5 + 0.1).bfloat16()
    x, y = xw[:, 32:32 + N1], yw[:, 8:8 + N2]                                  # views: row stride != width, 16-byte aligned starts
    ref = x.float().t() @ y.float()
    out = ops.linear_tn_f32out(x, y)
    scale = ref.abs().max().item() + 1e-6
    assert (out - ref).abs().max().item() <= 2e-5 * scale * max(1.0, (M / 64) ** 0.5), ((out - ref).abs().max().item(), scale)
    c0 = torch.randn(N1, N2 + 4, generator=g, device='cuda')
    c = c0.clone()
    ops.linear_tn_f32out(x, y, out=c[:, :N2], accumulate=True)
    assert torch.equal(c[:, N2:], c0[:, N2:])                                   # nothing written past the columns
    assert (c[:, :N2] - (c0[:, :N2] + ref)).abs().max().item() <= 3e-5 * (scale + c0.abs().max().item()) * max(1.0, (M / 64) ** 0.5)
    if True:                                                                   # the round-4 formulation: same products, fp32 accumulation in another order
        old = ops.linear_f32out(ops.transpose(x.contiguous(), 64), ops.transpose(y.contiguous(), 64))
        assert (out - old).abs().max().item() <= 2e-5 * scale * (M / 64) ** 0.5
    # deterministic: a second launch is bit-identical (also with the token loop split over work-groups: partial tiles added in a fixed order)
    assert torch.equal(out, ops.linear_tn_f32out(x, y))
    from arcflow_amd import _lib
    nws = _lib.load().afx_linear_tn_ws_bytes(M, N1, N2)
    if (M, N1, N2) in ((4608, 3072, 256), (4096, 256, 3072), (4608, 256, 15360)):
        assert nws > 0 and nws % (4 * N1 * N2) == 0 and nws // (4 * N1 * N2) >= 2         # the LoRA gradient shapes take the split
    if M <= 130:
        assert nws == 0                                                                    # too few K-steps to cut


@pytest.mark.parametrize('M,N,K,p', [(4608, 3072, 256, 0.05), (512, 12288, 256, 0.05), (130, 192, 64, 0.3), (4096, 15360, 256, 0.0)])
def test_linear_dropres_matches_product_then_masked_add(ops, M, N, K, p):
    """dx = dx0 + ((dy B) A) . keep/(1-p) in the GEMM's epilogue against the round-4 formulation (product to memory as bf16, then lora_dropout mode 3):
    the SAME mask bits (dropped positions keep the residual bit for bit), kept positions equal up to the bf16 rounding of the product that no longer happens;
    in place over the residual; every tile shape the launcher picks for these sizes."""
    g = torch.Generator(device='cuda').manual_seed(M + N)
    a = torch.randn(M, K, generator=g, device='cuda').bfloat16()
    w = (torch.randn(N, K, generator=g, device='cuda') * 0.1).bfloat16()
    res = torch.randn(M, N, generator=g, device='cuda').bfloat16()
    seed, row0 = 0x1234567, 4608
    old = res.clone()
    ops.lora_dropout(ops.linear(a, w), p, seed, row0, mode=3, out=old)
    new = res.clone()
    ops.linear_dropres(a, w, new, p, seed, row0, out=new)
    prod = a.float() @ w.float().t()
    keep = ops.lora_dropout(torch.ones(M, N, device='cuda', dtype=torch.bfloat16), p, seed, row0, mode=1) > 0
    ref = res.float() + torch.where(keep, prod / (1.0 - p), torch.zeros_like(prod))
    assert torch.equal(new[~keep], res[~keep])                                 # dropped: the residual, untouched
    assert abs(float((~keep).float().mean()) - p) < 0.01
    scale = ref.abs().max().item()
    assert (new.float() - ref).abs().max().item() <= 2 ** -8 * scale           # one bf16 rounding of the sum
    assert (new.float() - ref).abs().max().item() <= (old.float() - ref).abs().max().item() + 1e-6
    # under a kernel choice without the fused epilogue (A/B runs, parity tests: set_gemm_mode(2) = the 8-phase kernel) the call falls back to product + mask-and-add
    # instead of failing -- the round-4 formulation's bits exactly
    from arcflow_amd import _lib
    try:
        ops.set_gemm_mode(2)
        assert _lib.load().afx_gemm_dropres_available() == 0
        fb = res.clone()
        ops.linear_dropres(a, w, fb, p, seed, row0, out=fb)
        old2 = res.clone()
        ops.lora_dropout(ops.linear(a, w), p, seed, row0, mode=3, out=old2)
        assert torch.equal(fb, old2)
    finally:
        ops.set_gemm_mode(3)
    assert _lib.load().afx_gemm_dropres_available() == 1
