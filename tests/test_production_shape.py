"""Parity at the PRODUCTION shape (D = 3072, 24 heads x 128): the cases VERDICT r01 "What's weak" 1 asks for.

  * BASELINE.json configs[0] exactly: one FLUX double block (and one single block, one Qwen-Image block) at
    D=3072 / H=24 / 256 image + 77 text tokens against the fp32 CPU oracle (oracle/dit_ref.py restating
    diffusers' FluxTransformerBlock / FluxSingleTransformerBlock / QwenImageTransformerBlock,
    reference import sites lakonlab/models/architecture/arcflow/arcflux.py:9-13, arcqwen.py:9-11).
  * the same blocks at the full 1024^2 token count (4096 image + 512 text tokens = S 4608, and the ragged
    S = 4096 + 77 = 4173) -- the 3-heads-per-XCD mapping of the attention grid, the 18 x 12 ... 18 x 84 GEMM tile
    grids and the 7D fused single-block buffer, against the oracle.
  * attention alone at S = 4608 / 4224 / 4173, H = 24 against fp32 softmax(QK^T)V evaluated by torch on the device.
  * one full 19 + 38 block forward at 4096 + 512 tokens: finite, exp(logweights) sums to 1 over K, a second run is
    bit-identical, block 0's output equals a 1-block engine bound to the same tensors, and the LAST single block
    (rows 1 047 552.. of the stacked modulation matrix) equals the oracle's single block on the engine's own input.

Tolerances: bf16 trunk vs fp32 oracle on identical bf16-rounded weights, rel-L2 <= 1e-2 on the token matrix after
ONE block (output rounding of 4-6 chained bf16 GEMMs + bf16 P in attention), 2.5e-2 through the heads.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_BLOCK = 1e-2
TOL_HEAD = 2.5e-2


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp(min=1e-12)).item()


def _inputs(B, N, T, joint, pooled_dim, seed=1):
    g = torch.Generator().manual_seed(seed)
    hid = torch.randn(B, N, 64, generator=g).bfloat16()
    ctx = (torch.randn(B, T, joint, generator=g) * 0.5).bfloat16()
    pooled = (torch.randn(B, pooled_dim, generator=g) * 0.5).bfloat16() if pooled_dim else None
    return hid, ctx, pooled


def _flux_block_case(nd, ns, hp, wp, T, B=1, seed=3):
    """Engine vs oracle on a FLUX trunk of nd double + ns single blocks at full width; returns the errors."""
    from arcflow_amd import MMDiTEngine
    from oracle import dit_ref as D
    cfg = D.FluxCfg(num_layers=nd, num_single_layers=ns)          # defaults = FLUX.1-dev: 24 heads x 128, joint 4096
    assert cfg.dim == 3072 and cfg.heads == 24
    w = D.make_flux_weights(cfg, seed=seed)
    N = hp * wp
    hid, ctx, pooled = _inputs(B, N, T, cfg.joint_dim, cfg.pooled_dim, seed=seed + 1)
    t = torch.tensor([1.0, 0.7619][:B])
    gd = torch.full((B,), 3.5)
    wf = {k: v.float() for k, v in w.items()}
    ref_img = D.flux_forward(wf, cfg, hid.float(), ctx.float(), pooled.float(), t, gd, hp, wp, return_trunk=True)
    rm, rlw, rlg = D.flux_forward(wf, cfg, hid.float(), ctx.float(), pooled.float(), t, gd, hp, wp)
    eng = MMDiTEngine('flux', nd, ns)
    eng.load_state_dict(w)
    out = eng(hid.cuda(), t.cuda(), ctx.cuda(), pooled.cuda(), gd.cuda(), hp, wp)
    got_img = torch.empty(B * N, cfg.dim, dtype=torch.bfloat16, device='cuda')
    eng.export('x_final', got_img, B, N, T)
    torch.cuda.synchronize()
    return dict(trunk=rel_l2(got_img.view(B, N, -1).float(), ref_img), means=rel_l2(out.means.float(), rm),
                logg=rel_l2(out.loggammas.float(), rlg), logw=(out.logweights.float().cpu() - rlw).abs().max().item(),
                finite=bool(torch.isfinite(out.means.float()).all()))


@pytest.mark.parametrize('nd,ns', [(1, 0), (0, 1)])
def test_flux_block_configs0_shape(nd, ns):
    """BASELINE.json configs[0]: bs 1, 256 image + 77 text tokens, D 3072, 24 heads."""
    e = _flux_block_case(nd, ns, 16, 16, 77)
    assert e['finite']
    assert e['trunk'] < TOL_BLOCK, e
    assert e['means'] < TOL_HEAD and e['logg'] < TOL_HEAD and e['logw'] < 0.08, e


@pytest.mark.parametrize('nd,ns,T', [(1, 1, 512), (1, 1, 77)])
def test_flux_blocks_full_1024sq_token_count(nd, ns, T):
    """One double + one single block at 4096 image tokens (1024^2) with 512 (S = 4608 = 72 key tiles, 36 q-tiles x 24
    heads = 3 heads per XCD) or 77 text tokens (S = 4173: ragged last key tile and ragged last q-tile)."""
    e = _flux_block_case(nd, ns, 64, 64, T, seed=5)
    assert e['finite']
    assert e['trunk'] < TOL_BLOCK * 1.5, e                # two chained blocks
    assert e['means'] < TOL_HEAD and e['logg'] < TOL_HEAD and e['logw'] < 0.08, e


def test_flux_batch2_full_width():
    """B = 2 (per-sample modulation / gate slices, 4 problems per grouped GEMM launch) at full width."""
    e = _flux_block_case(1, 1, 16, 24, 77, B=2, seed=7)
    assert e['finite'] and e['trunk'] < TOL_BLOCK * 1.5 and e['means'] < TOL_HEAD, e


@pytest.mark.parametrize('hp,wp,T', [(16, 16, 77), (64, 64, 128)])
def test_qwen_block_full_width(hp, wp, T):
    from arcflow_amd import MMDiTEngine
    from oracle import dit_ref as D
    cfg = D.QwenCfg(num_layers=1)
    assert cfg.dim == 3072 and cfg.joint_dim == 3584
    w = D.make_qwen_weights(cfg, seed=11)
    N = hp * wp
    hid, ctx, _ = _inputs(1, N, T, cfg.joint_dim, 0, seed=12)
    t = torch.tensor([0.7619])
    wf = {k: v.float() for k, v in w.items()}
    rm, rlw, rlg = D.qwen_forward(wf, cfg, hid.float(), ctx.float(), t, hp, wp)
    eng = MMDiTEngine('qwen', 1, 0, joint_dim=cfg.joint_dim)
    eng.load_state_dict(w)
    out = eng(hid.cuda(), t.cuda(), ctx.cuda(), None, None, hp, wp)
    torch.cuda.synchronize()
    assert torch.isfinite(out.means.float()).all()
    assert rel_l2(out.means.float(), rm) < TOL_HEAD
    assert rel_l2(out.loggammas.float(), rlg) < TOL_HEAD
    assert (out.logweights.float().cpu() - rlw).abs().max().item() < 0.08


# ------------------------------------------------------------------------------------------ attention alone
@pytest.mark.parametrize('B,S,H', [(1, 4608, 24), (1, 4224, 24), (1, 4173, 24), (2, 1101, 24)])
def test_attention_production_shapes(B, S, H):
    """fp32 reference evaluated by torch on the device (2 GB score matrix per batch at S = 4608)."""
    from arcflow_amd import ops
    g = torch.Generator(device='cuda').manual_seed(S)
    q, k, v = (torch.randn(B, S, H, 128, generator=g, device='cuda').bfloat16() for _ in range(3))
    out = ops.attention(q, k, v)
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    ref = torch.empty(B, H, S, 128, device='cuda')
    for h0 in range(0, H, 4):                                   # 4 heads at a time keeps the fp32 scores at 340 MB
        s = torch.matmul(qf[:, h0:h0 + 4], kf[:, h0:h0 + 4].transpose(-1, -2)) * (128 ** -0.5)
        ref[:, h0:h0 + 4] = torch.matmul(torch.softmax(s, dim=-1), vf[:, h0:h0 + 4])
    ref = ref.transpose(1, 2).reshape(B, S, H * 128)
    assert torch.isfinite(out.float()).all()
    assert rel_l2(out, ref) < 1.2e-2
    # per-head errors: a wrong head -> XCD / q-tile mapping shows up as ONE bad head, not as a small global error
    per_head = ((out.float() - ref).view(B, S, H, 128).norm(dim=(1, 3)) / ref.view(B, S, H, 128).norm(dim=(1, 3)))
    assert per_head.max().item() < 1.5e-2, per_head


@pytest.mark.parametrize('B,S,H', [(1, 4608, 24), (1, 4224, 24), (1, 4133, 24), (2, 1101, 24)])
def test_attention_backward_production_shapes(B, S, H):
    """The generated backward streams (fused 864 + 864 work-group grid, list-scheduled XCD order, head -> XCD map: they only exist at H = 24) against the
    ANALYTIC fp32 backward of softmax attention evaluated by torch on the device, 4 heads at a time (at these shapes the backward
    had only been compared with the round-4 kernels).  dV = P^T dO, dP = dO V^T, dS = P (dP - rowsum(dO O)), dQ = dS K / sqrt(d), dK = dS^T Q / sqrt(d) --
    what SDPA's autograd computes under the reference's attention processor (arcflux.py:181-189).  Per-head bounds: a wrong head -> XCD map or
    work-group order shows up as ONE bad head, not as a small global error."""
    from arcflow_amd import ops
    g = torch.Generator(device='cuda').manual_seed(S + 7)
    q, k, v, do = (torch.randn(B, S, H, 128, generator=g, device='cuda').bfloat16() for _ in range(4))
    o, lse = ops.attention_fwd_lse(q, k, v)
    dq, dk, dv = ops.attention_bwd(q, k, v, o.reshape(B, S, H, 128), do, lse)
    torch.cuda.synchronize()
    qf, kf, vf, dof = (t.float().transpose(1, 2) for t in (q, k, v, do))          # [B, H, S, 128]
    rq, rk, rv = (torch.empty(B, H, S, 128, device='cuda') for _ in range(3))
    sc = 128 ** -0.5
    for h0 in range(0, H, 4):
        hs = slice(h0, h0 + 4)
        p = torch.softmax(torch.matmul(qf[:, hs], kf[:, hs].transpose(-1, -2)) * sc, dim=-1)
        of = torch.matmul(p, vf[:, hs])
        rv[:, hs] = torch.matmul(p.transpose(-1, -2), dof[:, hs])
        dp = torch.matmul(dof[:, hs], vf[:, hs].transpose(-1, -2))
        dp.sub_((dof[:, hs] * of).sum(-1, keepdim=True)).mul_(p).mul_(sc)          # dS / sqrt(d), in place (340 MB per buffer at S = 4608)
        rq[:, hs] = torch.matmul(dp, kf[:, hs])
        rk[:, hs] = torch.matmul(dp.transpose(-1, -2), qf[:, hs])
        del p, dp, of
    for name, got, ref, tol in (('dv', dv, rv, 1.5e-2), ('dq', dq, rq, 2e-2), ('dk', dk, rk, 2e-2)):
        got = got.float().reshape(B, S, H, 128).transpose(1, 2)
        assert torch.isfinite(got).all(), name
        err = ((got - ref).norm() / ref.norm()).item()
        assert err < tol, (name, err)
        per_head = (got - ref).norm(dim=(2, 3)) / ref.norm(dim=(2, 3))
        assert per_head.max().item() < 1.25 * tol, (name, per_head)


# ------------------------------------------------------------------------------------------ the whole network
def _unpack_single(P, i, D, nd):
    """diffusers-keyed fp32 weights of single block i from the packed engine tensors (inverse of weights.pack_flux)."""
    p = f'single_transformer_blocks.{i}.'
    fw, fb = P[f's{i}.fused.weight'].float().cpu(), P[f's{i}.fused.bias'].float().cpu()
    w = {}
    for j, nm in enumerate(('attn.to_k', 'attn.to_v', 'attn.to_q')):
        w[p + nm + '.weight'], w[p + nm + '.bias'] = fw[j * D:(j + 1) * D], fb[j * D:(j + 1) * D]
    w[p + 'proj_mlp.weight'], w[p + 'proj_mlp.bias'] = fw[3 * D:], fb[3 * D:]
    w[p + 'proj_out.weight'], w[p + 'proj_out.bias'] = P[f's{i}.out.weight'].float().cpu(), P[f's{i}.out.bias'].float().cpu()
    qk = P[f's{i}.qknorm'].float().cpu()
    w[p + 'attn.norm_q.weight'], w[p + 'attn.norm_k.weight'] = qk[0], qk[1]
    r0 = (nd * 12 + i * 3) * D
    w[p + 'norm.linear.weight'] = P['mod.weight'][r0:r0 + 3 * D].float().cpu()
    w[p + 'norm.linear.bias'] = P['mod.bias'][r0:r0 + 3 * D].float().cpu()
    return w


def test_full_flux12b_forward_properties_and_block_parity():
    from arcflow_amd import MMDiTEngine
    from arcflow_amd.weights import random_packed
    from oracle import dit_ref as D
    nd, ns, Dm, N, T, hp = 19, 38, 3072, 4096, 512, 64
    S = N + T
    P = random_packed('flux', nd, ns, 'cuda', seed=0)
    assert P['mod.weight'].shape == (1056768, Dm)
    eng = MMDiTEngine('flux', nd, ns)
    eng.bind_packed(P)
    g = torch.Generator(device='cuda').manual_seed(42)
    x = torch.randn(1, N, 64, generator=g, device='cuda').bfloat16()
    ctx = (torch.randn(1, T, 4096, generator=g, device='cuda') * 0.1).bfloat16()
    pooled = (torch.randn(1, 768, generator=g, device='cuda') * 0.1).bfloat16()
    t, gd = torch.tensor([0.7619], device='cuda'), torch.full((1,), 3.5, device='cuda')
    ck = torch.zeros(nd + ns, S, Dm, dtype=torch.bfloat16, device='cuda')
    eng.set_checkpoint_buffer(ck)
    out = eng(x, t, ctx, pooled, gd, hp, hp)
    x_last = torch.empty(S, Dm, dtype=torch.bfloat16, device='cuda')
    eng.export('x_tokens', x_last, 1, N, T)
    temb = torch.empty(1, Dm, device='cuda')
    eng.export('temb', temb, 1, N, T)
    eng.set_checkpoint_buffer(None)
    torch.cuda.synchronize()
    # ---- properties --------------------------------------------------------------------------------------------
    for k in ('means', 'logweights', 'loggammas'):
        assert torch.isfinite(out[k].float()).all(), k
    assert out.means.shape == (1, N, 16, 64) and out.logweights.shape == (1, N, 16, 4) and out.loggammas.shape == (1, N, 15, 4)
    assert torch.allclose(out.logweights.float().exp().sum(dim=2), torch.ones(1, N, 4, device='cuda'), atol=2e-2)
    out2 = eng(x, t, ctx, pooled, gd, hp, hp)
    torch.cuda.synchronize()
    for k in ('means', 'logweights', 'loggammas'):
        assert torch.equal(out[k], out2[k]), f'{k}: second run differs'
    assert torch.isfinite(ck.float()).all()
    # ---- block 0 of the 57-block engine == a 1-block engine on the same tensors (bit-exact) ---------------------------
    sub = {k: v for k, v in P.items() if k.startswith(('x_in', 'ctx_in', 'temb.', 'd0.', 'head'))}
    sub['mod.weight'] = torch.cat([P['mod.weight'][:12 * Dm], P['mod.weight'][-2 * Dm:]]).contiguous()
    sub['mod.bias'] = torch.cat([P['mod.bias'][:12 * Dm], P['mod.bias'][-2 * Dm:]]).contiguous()
    e1 = MMDiTEngine('flux', 1, 0)
    e1.bind_packed(sub)
    e1(x, t, ctx, pooled, gd, hp, hp)
    x1 = torch.empty(S, Dm, dtype=torch.bfloat16, device='cuda')
    e1.export('x_tokens', x1, 1, N, T)
    torch.cuda.synchronize()
    assert torch.equal(x1, ck[1]), 'block 0 output differs between the 57-block and the 1-block launch plan'
    # ---- the LAST single block (modulation rows 1 047 552 ..) vs the oracle on the engine's own block input -------------
    i = ns - 1
    w = _unpack_single(P, i, Dm, nd)
    cos, sin = D.flux_rope_tables(hp, hp, T)
    xin = ck[nd + i].float().cpu()[None]
    ref = D.flux_single_block(w, f'single_transformer_blocks.{i}.', D.FluxCfg(), xin, temb.cpu(), cos, sin)
    err = rel_l2(x_last.float()[None], ref)
    assert err < TOL_BLOCK, err
    # the increment itself (residual removed): a dead block would pass the check above
    inc = rel_l2(x_last.float().cpu()[None] - xin, ref - xin)
    assert inc < 8e-2, inc


@pytest.mark.parametrize('family,hp,T,B', [('flux', 32, 128, 2), ('qwen', 32, 128, 2), ('flux', 16, 77, 1)])
def test_fp8_block_scaled_forward_at_full_width(family, hp, T, B, monkeypatch):
    """BASELINE.json configs[4]'s forward format at the production width: e4m3 operands with one E8M0 scale per row and 128 columns, written by
    the LayerNorm-modulate kernel, by the mlp / k|v|q|mlp GEMMs' epilogues and (attention output) by the quantiser -- no bf16 copy of the mlp
    hidden exists.  One double (+ one single) block at 1024 + 128 tokens, two samples: against the bf16 engine, and against the same engine with one
    scale per row and a quantisation pass per GEMM (AFX_FP8_MX=0).  Stated tolerance of the fp8 mode: 8e-2, as for the row-scaled path (tests/test_hip_engine.py)."""
    from arcflow_amd import MMDiTEngine
    from oracle import dit_ref as D
    wp = hp                  # (16 x 16 + 77 tokens, one sample = BASELINE configs[0]: 333 rows -- no fused LayerNorm kernel, launches of a few tiles, ragged S)
    N = hp * wp
    if family == 'flux':
        cfg = D.FluxCfg(num_layers=1, num_single_layers=1)
        w = D.make_flux_weights(cfg, seed=3)
        nd, ns = 1, 1
    else:
        cfg = D.QwenCfg(num_layers=2)
        w = D.make_qwen_weights(cfg, seed=3)
        nd, ns = 2, 0
    hid, ctx, pooled = _inputs(B, N, T, cfg.joint_dim, cfg.pooled_dim if family == 'flux' else 0, seed=4)
    t = torch.tensor([1.0, 0.7619][:B]).cuda()
    gd = torch.full((B,), 3.5).cuda() if family == 'flux' else None
    res = {}
    for mode in ('bf16', 'mx', 'row'):
        monkeypatch.setenv('AFX_FP8_MX', '0' if mode == 'row' else '1')
        eng = MMDiTEngine(family, nd, ns, joint_dim=cfg.joint_dim)
        eng.load_state_dict(w)
        if mode != 'bf16':
            eng.enable_fp8()
        out = eng(hid.cuda(), t, ctx.cuda(), None if pooled is None else pooled.cuda(), gd, hp, wp)
        xf = torch.empty(B * N, cfg.dim, dtype=torch.bfloat16, device='cuda')
        eng.export('x_final', xf, B, N, T)
        torch.cuda.synchronize()
        res[mode] = (xf.float().clone(), out.means.float().clone())
    for i, name in enumerate(('trunk', 'means')):
        ref = res['bf16'][i]
        e_mx, e_row = rel_l2(res['mx'][i], ref), rel_l2(res['row'][i], ref)
        assert 0 < e_mx < 8e-2 and 0 < e_row < 8e-2, (name, e_mx, e_row)
        assert e_mx < 1.25 * e_row, (name, e_mx, e_row)
        assert not torch.equal(res['mx'][i], res['row'][i])
        print(family, name, 'block-scaled', e_mx, 'row-scaled', e_row)
