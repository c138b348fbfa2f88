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
