"""AutoencoderKL decoder (HIP implicit-GEMM convolutions) against the fp32 CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_conv3x3_implicit_gemm_vs_torch():
    import ctypes as C
    from arcflow_amd import _lib
    from arcflow_amd.vae import _Grid, _p, _s
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    H, W, ci, co = 9, 13, 64, 72
    x = torch.randn(1, ci, H, W, generator=g).bfloat16()
    wt = (torch.randn(co, ci, 3, 3, generator=g) * 0.05).bfloat16()
    b = torch.randn(co, generator=g).bfloat16()
    res = torch.randn(1, co, H, W, generator=g).bfloat16()
    gx, gr, gy = _Grid(H, W, ci, 'cuda'), _Grid(H, W, co, 'cuda'), _Grid(H, W, co, 'cuda')
    gx.t.view(H + 2, W + 2, ci)[1:-1, 1:-1] = x[0].permute(1, 2, 0).cuda()
    gr.t.view(H + 2, W + 2, co)[1:-1, 1:-1] = res[0].permute(1, 2, 0).cuda()
    wp = wt.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous().cuda()
    _lib.check(lib.afx_conv3x3_bf16(_p(gx.t), _p(wp), _p(b.cuda()), _p(gy.t), H, W, ci, co, _p(gr.t), _s()))
    out = gy.t.view(H + 2, W + 2, co)
    ref = torch.nn.functional.conv2d(x.float(), wt.float(), b.float(), padding=1) + res.float()
    got = out[1:-1, 1:-1].permute(2, 0, 1).float().cpu()
    assert ((got - ref[0]).norm() / ref.norm()).item() < 5e-3
    border = torch.cat([out[0].flatten(), out[-1].flatten(), out[:, 0].flatten(), out[:, -1].flatten()])
    assert border.abs().max().item() == 0          # the epilogue keeps the padded grid's border zero


@pytest.mark.parametrize('H,W,ci,co,groups,with_res', [(9, 13, 64, 128, 32, True), (20, 17, 128, 128, 32, False), (6, 40, 64, 64, 8, True),
                                                      (31, 33, 64, 128, 16, False), (300, 70, 64, 128, 32, True)])
def test_conv3x3_epilogue_groupnorm_sums(H, W, ci, co, groups, with_res):
    """afx_conv3x3_bf16_stats: the GroupNorm sums of the convolution's OUTPUT grid come out of the GEMM epilogue (slotted fp64 partial sums) and
    afx_groupnorm_nhwc_from_stats normalises with them -- against sums of the stored grid and against afx_groupnorm_nhwc (its own statistics
    pass) on the same grid.  Channels per group 4 / 4 / 8 / 8 / 4 (Cout <= 128: the 256x128-tile kernel)."""
    from arcflow_amd import _lib
    from arcflow_amd.vae import _Grid, _p, _s
    lib = _lib.load()
    assert lib.afx_conv_stats_available() == 1
    g = torch.Generator().manual_seed(H * W + co)
    x = torch.randn(ci, H, W, generator=g).bfloat16()
    wt = (torch.randn(co, ci, 3, 3, generator=g) * 0.05).bfloat16()
    b = torch.randn(co, generator=g).bfloat16()
    gx, gr, gy = _Grid(H, W, ci, 'cuda'), _Grid(H, W, co, 'cuda'), _Grid(H, W, co, 'cuda')
    gx.t.view(H + 2, W + 2, ci)[1:-1, 1:-1] = x.permute(1, 2, 0).cuda()
    gr.t.view(H + 2, W + 2, co)[1:-1, 1:-1] = torch.randn(H, W, co, generator=g).bfloat16().cuda()
    wp = wt.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous().cuda()
    slots = torch.full((64, groups, 2), 7.0, dtype=torch.float64, device='cuda')          # the call zeroes it
    _lib.check(lib.afx_conv3x3_bf16_stats(_p(gx.t), _p(wp), _p(b.cuda()), _p(gy.t), H, W, ci, co, _p(gr.t) if with_res else None,
                                          _p(slots), groups, _s()))
    got = slots.sum(0).cpu()                                                               # [groups, 2]
    y = gy.t.view(H + 2, W + 2, co).double().cpu()
    assert y[0].abs().max() == 0 and y[:, 0].abs().max() == 0                              # border still zero
    yg = y.view(-1, groups, co // groups)
    ref = torch.stack([yg.sum((0, 2)), (yg * yg).sum((0, 2))], 1)
    # the epilogue sums the fp32 values BEFORE their rounding to bf16 (2^-9 relative per element): a random walk of n such errors
    n = (H + 2) * (W + 2) * (co // groups)
    assert ((got[:, 0] - ref[:, 0]).abs() <= 2.0 ** -8 * (n * ref[:, 1]).sqrt()).all(), (got[:, 0] - ref[:, 0]).abs().max().item()
    assert ((got[:, 1] - ref[:, 1]).abs() <= 2e-3 * ref[:, 1]).all(), ((got[:, 1] - ref[:, 1]).abs() / ref[:, 1]).max().item()
    gamma, beta = torch.randn(co, generator=g).cuda(), torch.randn(co, generator=g).cuda()
    ws = torch.zeros(lib.afx_groupnorm_ws_bytes(co, groups) // 8, dtype=torch.float64, device='cuda')
    assert lib.afx_groupnorm_ws_bytes(co, groups) == 8 * ((2 + 2 * 64) * groups + co)
    ya, yb = _Grid(H, W, co, 'cuda'), _Grid(H, W, co, 'cuda')
    _lib.check(lib.afx_groupnorm_nhwc_from_stats(_p(gy.t), _p(ya.t), _p(slots), _p(ws), H, W, co, groups, _p(gamma), _p(beta), 1e-6, 1, _s()))
    _lib.check(lib.afx_groupnorm_nhwc(_p(gy.t), _p(yb.t), _p(ws), H, W, co, groups, _p(gamma), _p(beta), 1e-6, 1, _s()))
    d = (ya.t.float() - yb.t.float()).abs().max().item()
    assert d <= 2.0 ** -6 * yb.t.float().abs().max().item(), d


@pytest.mark.parametrize('hp,wp', [(4, 4), (3, 5)])
def test_decoder_vs_oracle(hp, wp):
    from arcflow_amd.vae import AutoencoderKLDecoder
    from oracle import arcflow_ref as R
    from oracle import vae_ref as V
    chans = (64, 128, 128, 128)
    w = V.make_decoder_weights(chans, seed=1)
    g = torch.Generator().manual_seed(2)
    tok = torch.randn(1, hp * wp, 64, generator=g)
    dec = AutoencoderKLDecoder(w, chans, norm_num_groups=16)
    img = dec.decode_packed(tok.cuda(), hp, wp)
    z = R.unpack_latents(tok, hp, wp) / 0.3611 + 0.1159
    ref = V.decode(w, z.bfloat16().float(), chans, groups=16)
    assert img.shape == ref.shape == (1, 3, 16 * hp, 16 * wp)
    rel = ((img.cpu() - ref).norm() / ref.norm()).item()
    assert rel < 3e-2, rel


def test_pipeline_decodes_images_end_to_end():
    """pipe(..., output_type='pt'): 2-NFE loop + HIP VAE decode vs the CPU oracle chain."""
    from arcflow_amd import FlowMatchEulerDiscreteScheduler
    from arcflow_amd.pipelines import ArcFluxPipeline
    from arcflow_amd.vae import AutoencoderKLDecoder
    from oracle import arcflow_ref as R
    from oracle import dit_ref as D
    from oracle import vae_ref as V
    cfg = D.FluxCfg(num_layers=1, num_single_layers=1, heads=2, joint_dim=128, pooled_dim=64)
    w = D.make_flux_weights(cfg, seed=9)
    tcfg = dict(num_layers=1, num_single_layers=1, num_attention_heads=2, attention_head_dim=128, in_channels=64,
                joint_attention_dim=128, pooled_projection_dim=64, guidance_embeds=True)
    pipe = ArcFluxPipeline.from_state_dict(tcfg, w, scheduler=FlowMatchEulerDiscreteScheduler(shift=3.2))
    chans = (64, 128, 128, 128)
    vw = V.make_decoder_weights(chans, seed=3)
    pipe.vae = AutoencoderKLDecoder(vw, chans, norm_num_groups=16)
    g = torch.Generator().manual_seed(4)
    pe = (torch.randn(1, 12, 128, generator=g) * 0.5).bfloat16()
    pp = (torch.randn(1, 64, generator=g) * 0.5).bfloat16()
    lat = torch.randn(1, 16, 64, generator=g)
    img = pipe(prompt_embeds=pe, pooled_prompt_embeds=pp, latents=lat, width=64, height=64, num_inference_steps=2,
               timestep_ratio=1.0, output_type='pt').images
    assert img.shape == (1, 3, 64, 64)
    x = lat.clone()
    sig, _ = R.inference_sigmas(2)
    for i in range(2):
        m, lw, lg = D.flux_forward(w, cfg, x.bfloat16().float(), pe.float(), pp.float(), torch.tensor([sig[i]]), torch.tensor([3.5]), 4, 4)
        x = R.momentum_step_packed(x, m, lw, lg, sig[i], sig[i], sig[i + 1])
    ref = V.decode(vw, (R.unpack_latents(x, 4, 4) / 0.3611 + 0.1159).bfloat16().float(), chans, groups=16)
    assert ((img.cpu().float() - ref).norm() / ref.norm()).item() < 4e-2
    pil = pipe(prompt_embeds=pe, pooled_prompt_embeds=pp, latents=lat, width=64, height=64, num_inference_steps=2,
               timestep_ratio=1.0).images[0]
    assert pil.size == (64, 64)


def test_rmsnorm_nhwc_padded_channels():
    import ctypes as C
    from arcflow_amd import _lib
    from arcflow_amd.vae import _p, _s
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    rows, creal, cpad = 37, 96, 128
    x = torch.zeros(rows, cpad)
    x[:, :creal] = torch.randn(rows, creal, generator=g)
    x[5] = 0                                           # a border row stays exactly zero
    gamma = torch.zeros(cpad)
    gamma[:creal] = 1 + 0.1 * torch.randn(creal, generator=g)
    xb = x.bfloat16().cuda()
    for act in (0, 1):
        y = torch.empty_like(xb)
        _lib.check(lib.afx_rmsnorm_nhwc(_p(xb), _p(y), rows, cpad, creal, _p(gamma.cuda()), act, _s()))
        ref = torch.nn.functional.normalize(xb.float().cpu()[:, :creal], dim=1) * creal ** 0.5 * gamma[:creal]
        ref = torch.nn.functional.silu(ref) if act else ref
        got = y.float().cpu()
        assert (got[:, :creal] - ref).abs().max().item() < 2e-2
        assert got[:, creal:].abs().max().item() == 0 and got[5].abs().max().item() == 0


@pytest.mark.parametrize('hp,wp,dim', [(4, 4, 32), (3, 5, 96)])
def test_qwen_decoder_vs_oracle(hp, wp, dim):
    """dim=96 is the released width (384/192/96 channels: the 96-channel stage runs on grids padded to 128)."""
    from arcflow_amd.vae import AutoencoderKLQwenImageDecoder
    from oracle import arcflow_ref as R
    from oracle import vae_qwen_ref as V
    w = V.make_decoder_weights(dim=dim, seed=1)
    g = torch.Generator().manual_seed(2)
    mean = (torch.randn(16, generator=g) * 0.3).tolist()
    std = (1.0 + 0.5 * torch.rand(16, generator=g)).tolist()
    tok = torch.randn(1, hp * wp, 64, generator=g)
    dec = AutoencoderKLQwenImageDecoder(w, mean, std)
    img = dec.decode_packed(tok.cuda(), hp, wp)
    z = R.unpack_latents(tok, hp, wp) * torch.tensor(std).view(1, 16, 1, 1) + torch.tensor(mean).view(1, 16, 1, 1)
    ref = V.decode(w, z)
    assert img.shape == ref.shape == (1, 3, 16 * hp, 16 * wp)
    rel = ((img.cpu() - ref).norm() / ref.norm()).item()
    assert rel < 3e-2, rel
    assert img.abs().max().item() <= 1.0


def test_qwen_pipeline_decodes_images_end_to_end():
    from arcflow_amd import FlowMatchEulerDiscreteScheduler
    from arcflow_amd.pipelines import ArcQwenImagePipeline
    from arcflow_amd.vae import AutoencoderKLQwenImageDecoder
    from oracle import dit_ref as D
    from oracle import vae_qwen_ref as V
    cfg = D.QwenCfg(num_layers=2, heads=2, joint_dim=128)
    w = D.make_qwen_weights(cfg, seed=9)
    tcfg = dict(num_layers=2, num_attention_heads=2, attention_head_dim=128, in_channels=64, joint_attention_dim=128)
    pipe = ArcQwenImagePipeline.from_state_dict(tcfg, w, scheduler=FlowMatchEulerDiscreteScheduler(shift=3.2))
    vw = V.make_decoder_weights(dim=32, seed=3)
    g = torch.Generator().manual_seed(4)
    mean, std = (torch.randn(16, generator=g) * 0.3).tolist(), (1.0 + 0.5 * torch.rand(16, generator=g)).tolist()
    pipe.vae = AutoencoderKLQwenImageDecoder(vw, mean, std)
    pe = (torch.randn(1, 12, 128, generator=g) * 0.5).bfloat16()
    lat = torch.randn(1, 16, 64, generator=g)
    out_lat = pipe(prompt_embeds=pe, prompt_embeds_mask=torch.ones(1, 12, dtype=torch.long), latents=lat, width=64, height=64,
                   num_inference_steps=2, timestep_ratio=1.0, output_type='latent').images
    img = pipe(prompt_embeds=pe, prompt_embeds_mask=torch.ones(1, 12, dtype=torch.long), latents=lat, width=64, height=64,
               num_inference_steps=2, timestep_ratio=1.0, output_type='pt').images
    from oracle import arcflow_ref as R
    z = R.unpack_latents(out_lat.float().cpu(), 4, 4) * torch.tensor(std).view(1, 16, 1, 1) + torch.tensor(mean).view(1, 16, 1, 1)
    ref = V.decode(vw, z)
    assert img.shape == (1, 3, 64, 64)
    assert ((img.float().cpu() - ref).norm() / ref.norm()).item() < 3e-2


# ------------------------------------------------------------------------------------------ released sizes, 1024^2
def _on_device(w):
    return {k: v.cuda() for k, v in w.items()}


def test_flux_decoder_1024sq_released_width_vs_oracle_on_device():
    """The released AutoencoderKL decoder width (128/256/512/512, 32 groups) at 1024 x 1024: the 128-channel full-resolution
    stage (half-filled 256-wide GEMM tiles) and conv_out (3 channels) that the 64-pixel tests never reach.  The fp32 oracle
    (oracle/vae_ref.py, torch ops) is evaluated on the device here -- on the host it would need minutes."""
    from arcflow_amd.vae import AutoencoderKLDecoder
    from oracle import arcflow_ref as R
    from oracle import vae_ref as V
    chans = (128, 256, 512, 512)
    w = V.make_decoder_weights(chans, seed=5)
    g = torch.Generator().manual_seed(6)
    hp = wp = 64
    tok = torch.randn(1, hp * wp, 64, generator=g)
    dec = AutoencoderKLDecoder(w, chans, norm_num_groups=32)
    img = dec.decode_packed(tok.cuda(), hp, wp)
    z = R.unpack_latents(tok, hp, wp) / 0.3611 + 0.1159
    with torch.no_grad():
        ref = V.decode(_on_device(w), z.bfloat16().float().cuda(), chans, groups=32)
    assert img.shape == ref.shape == (1, 3, 1024, 1024)
    assert torch.isfinite(img.float()).all()
    rel = ((img.float() - ref).norm() / ref.norm()).item()
    assert rel < 3e-2, rel
    # no stripe / tile of the image is off on its own (a mis-addressed tile hides inside a global norm)
    d = (img.float() - ref).reshape(3, 16, 64, 16, 64).pow(2).sum(dim=(0, 2, 4)).sqrt()
    r = ref.reshape(3, 16, 64, 16, 64).pow(2).sum(dim=(0, 2, 4)).sqrt()
    assert (d / r).max().item() < 6e-2, (d / r).max().item()


def test_qwen_decoder_1024sq_released_width_vs_oracle_on_device():
    from arcflow_amd.vae import AutoencoderKLQwenImageDecoder
    from oracle import arcflow_ref as R
    from oracle import vae_qwen_ref as V
    w = V.make_decoder_weights(dim=96, seed=7)
    g = torch.Generator().manual_seed(8)
    mean = (torch.randn(16, generator=g) * 0.3).tolist()
    std = (1.0 + 0.5 * torch.rand(16, generator=g)).tolist()
    hp = wp = 64
    tok = torch.randn(1, hp * wp, 64, generator=g)
    dec = AutoencoderKLQwenImageDecoder(w, mean, std)
    img = dec.decode_packed(tok.cuda(), hp, wp)
    z = R.unpack_latents(tok, hp, wp) * torch.tensor(std).view(1, 16, 1, 1) + torch.tensor(mean).view(1, 16, 1, 1)
    with torch.no_grad():
        ref = V.decode(_on_device(w), z.cuda())
    assert img.shape == ref.shape == (1, 3, 1024, 1024)
    rel = ((img.float() - ref).norm() / ref.norm()).item()
    assert rel < 3e-2, rel
    d = (img.float() - ref).reshape(3, 16, 64, 16, 64).pow(2).sum(dim=(0, 2, 4)).sqrt()
    r = ref.reshape(3, 16, 64, 16, 64).pow(2).sum(dim=(0, 2, 4)).sqrt()
    assert (d / r).max().item() < 6e-2, (d / r).max().item()


@pytest.mark.parametrize('rows,cols', [(7, 64), (5, 1000), (3, 16384), (4, 1023), (2, 20000)])
def test_softmax_rows_both_kernels(rows, cols):
    """afx_softmax_rows_f32 (mid-block attention of the AutoencoderKL decoder): the single-read register kernel (cols % 4 == 0, <= 16384)
    and the three-pass fallback against torch.softmax; bf16 output: 2^-8 relative."""
    from arcflow_amd import _lib
    from arcflow_amd.vae import _p, _s
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows * cols)
    s = (torch.randn(rows, cols, generator=g) * 4).cuda()
    p = torch.empty(rows, cols, dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.afx_softmax_rows_f32(_p(s), cols, _p(p), cols, rows, cols, 0.37, _s()))
    ref = torch.softmax(s.double() * 0.37, -1)
    err = ((p.double() - ref).abs() / ref.clamp(min=1e-30)).max().item()
    assert err < 2.0 ** -7, err
    assert abs(p.double().sum(-1) - 1).max().item() < 2e-3


@pytest.mark.parametrize('H,W,ci,co', [(5, 7, 64, 72), (16, 9, 128, 256), (33, 20, 64, 128)])
def test_upsample_folded_into_conv_vs_torch(H, W, ci, co):
    """afx_upconv3x3_bf16 = conv3x3(nearest-2x upsample(x)) as four 2x2 phase convolutions on the low-resolution grid (diffusers Upsample2D:
    F.interpolate(scale_factor=2, mode='nearest') then conv): against torch on the same bf16 inputs; the 2x grid's border stays zero and every
    interior pixel is written (the output buffer starts as NaN)."""
    from arcflow_amd import _lib
    from arcflow_amd.vae import _Grid, _p, _s, phase_weights
    lib = _lib.load()
    g = torch.Generator().manual_seed(H * W + co)
    x = torch.randn(1, ci, H, W, generator=g).bfloat16()
    wt = (torch.randn(co, ci, 3, 3, generator=g) * 0.05).bfloat16()
    b = torch.randn(co, generator=g).bfloat16()
    gx, gy = _Grid(H, W, ci, 'cuda'), _Grid(2 * H, 2 * W, co, 'cuda')
    gx.t.view(H + 2, W + 2, ci)[1:-1, 1:-1] = x[0].permute(1, 2, 0).cuda()
    gy.t.fill_(float('nan'))
    w9 = wt.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous().cuda()
    w4 = phase_weights(w9, ci)
    assert w4.shape == (4, co, 4 * ci)
    _lib.check(lib.afx_upconv3x3_bf16(_p(gx.t), _p(w4), _p(b.cuda()), _p(gy.t), H, W, ci, co, _s()))
    out = gy.t.view(2 * H + 2, 2 * W + 2, co)
    assert torch.isfinite(out.float()).all()                                   # every position of the 2x grid was written
    up = torch.nn.functional.interpolate(x.float(), scale_factor=2, mode='nearest')
    ref = torch.nn.functional.conv2d(up, wt.float(), b.float(), padding=1)[0]
    got = out[1:-1, 1:-1].permute(2, 0, 1).float().cpu()
    # the phase kernels are sums of up to four bf16 taps rounded once more to bf16: 2^-8 relative on the weights
    assert ((got - ref).norm() / ref.norm()).item() < 8e-3, ((got - ref).norm() / ref.norm()).item()
    border = torch.cat([out[0].flatten(), out[-1].flatten(), out[:, 0].flatten(), out[:, -1].flatten()])
    assert border.abs().max().item() == 0
