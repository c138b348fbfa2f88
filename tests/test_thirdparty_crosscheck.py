"""Opportunistic pin of the third-party oracles (VERDICT r2 "missing 4"): wherever `diffusers` / `peft` /
`bitsandbytes` happen to be importable, the restatements in oracle/ are compared with the REAL modules on the same weights
and inputs (fp32, CPU; the bitsandbytes check needs its GPU build).  Where the dependency is absent the test SKIPS with the reason
in the log -- in the build container and on the GPU pool none of the three is installed, so these rows stay "parity unpinned"
there; the tests cost nothing and pin themselves the day the dependency exists.  The reference pins diffusers==0.35.1 and
peft==0.17.0 (requirements.txt:4-5); block classes, constructor arguments and call signatures below follow those versions."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TOL = 2e-4          # fp32 vs fp32: summation order only


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-12)).item()


def _load(module, w, prefix):
    """copy oracle-keyed fp32 weights `prefix...` into a diffusers module; every parameter must be covered"""
    sd = {k[len(prefix):]: v.float() for k, v in w.items() if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not missing, missing
    return module.float().eval()


# ------------------------------------------------------------------------------------------------------------- diffusers: MMDiT blocks
def test_flux_double_block_vs_diffusers():
    tf = pytest.importorskip('diffusers.models.transformers.transformer_flux', reason='diffusers not installed: dit_ref.flux_double_block stays unpinned')
    from oracle import dit_ref as D
    cfg = D.FluxCfg(num_layers=1, num_single_layers=0, heads=2, joint_dim=128, pooled_dim=64)
    w = {k: v.float() for k, v in D.make_flux_weights(cfg, seed=0).items()}
    g = torch.Generator().manual_seed(1)
    hp, wp, T = 4, 6, 9
    img, txt, temb = torch.randn(2, hp * wp, cfg.dim, generator=g), torch.randn(2, T, cfg.dim, generator=g), torch.randn(2, cfg.dim, generator=g)
    cos, sin = D.flux_rope_tables(hp, wp, T, cfg.axes_dims, bf16_round=False)
    ref_txt, ref_img = D.flux_double_block(w, 'transformer_blocks.0.', cfg, img, txt, temb, cos, sin)
    blk = _load(tf.FluxTransformerBlock(dim=cfg.dim, num_attention_heads=cfg.heads, attention_head_dim=cfg.head_dim), w, 'transformer_blocks.0.')
    pos = tf.FluxPosEmbed(theta=10000, axes_dim=list(cfg.axes_dims))
    rot = pos(D.flux_ids(hp, wp, T))                     # (cos, sin) each [S, 128], pairs repeated (repeat_interleave_real)
    assert _rel(rot[0][:, 0::2], cos) < 1e-6 and _rel(rot[1][:, 0::2], sin) < 1e-6
    with torch.no_grad():
        out_txt, out_img = blk(hidden_states=img, encoder_hidden_states=txt, temb=temb, image_rotary_emb=rot)
    assert _rel(ref_img, out_img) < TOL and _rel(ref_txt, out_txt) < TOL


def test_flux_single_block_vs_diffusers():
    tf = pytest.importorskip('diffusers.models.transformers.transformer_flux', reason='diffusers not installed: dit_ref.flux_single_block stays unpinned')
    from oracle import dit_ref as D
    cfg = D.FluxCfg(num_layers=0, num_single_layers=1, heads=2, joint_dim=128, pooled_dim=64)
    w = {k: v.float() for k, v in D.make_flux_weights(cfg, seed=2).items()}
    g = torch.Generator().manual_seed(3)
    hp, wp, T = 4, 4, 7
    x, temb = torch.randn(1, T + hp * wp, cfg.dim, generator=g), torch.randn(1, cfg.dim, generator=g)
    cos, sin = D.flux_rope_tables(hp, wp, T, cfg.axes_dims, bf16_round=False)
    ref = D.flux_single_block(w, 'single_transformer_blocks.0.', cfg, x, temb, cos, sin)
    blk = _load(tf.FluxSingleTransformerBlock(dim=cfg.dim, num_attention_heads=cfg.heads, attention_head_dim=cfg.head_dim, mlp_ratio=4.0), w,
                'single_transformer_blocks.0.')
    rot = tf.FluxPosEmbed(theta=10000, axes_dim=list(cfg.axes_dims))(D.flux_ids(hp, wp, T))
    with torch.no_grad():            # diffusers 0.35: (hidden_states [image], encoder_hidden_states [text], temb, rope) -> (text, image)
        out = blk(hidden_states=x[:, T:], encoder_hidden_states=x[:, :T], temb=temb, image_rotary_emb=rot)
    out = torch.cat(list(out), dim=1) if isinstance(out, (tuple, list)) else out
    assert _rel(ref, out) < TOL


def test_flux_embedders_and_norm_out_vs_diffusers():
    emb = pytest.importorskip('diffusers.models.embeddings', reason='diffusers not installed: dit_ref embedders stay unpinned')
    norm = pytest.importorskip('diffusers.models.normalization')
    from oracle import dit_ref as D
    cfg = D.FluxCfg(num_layers=0, num_single_layers=0, heads=2, joint_dim=128, pooled_dim=64)
    w = {k: v.float() for k, v in D.make_flux_weights(cfg, seed=4).items()}
    g = torch.Generator().manual_seed(5)
    t, gd, pooled = torch.rand(3, generator=g), torch.full((3,), 3.5), torch.randn(3, 64, generator=g)
    ref = D.flux_temb(w, cfg, t, gd, pooled)
    m = _load(emb.CombinedTimestepGuidanceTextProjEmbeddings(embedding_dim=cfg.dim, pooled_projection_dim=64), w, 'time_text_embed.')
    with torch.no_grad():
        out = m(D.cond_cast(D.cond_cast(t) * 1000), D.cond_cast(D.cond_cast(gd) * 1000), pooled)   # arcflux.py:160-162
    assert _rel(ref, out) < TOL
    x = torch.randn(3, 10, cfg.dim, generator=g)
    no = _load(norm.AdaLayerNormContinuous(cfg.dim, cfg.dim, elementwise_affine=False, eps=1e-6), w, 'norm_out.')
    sc, sh = D.lin(w, 'norm_out.linear', torch.nn.functional.silu(ref)).chunk(2, dim=1)         # scale first
    with torch.no_grad():
        assert _rel(D.layer_norm(x) * (1 + sc[:, None]) + sh[:, None], no(x, ref)) < TOL


def test_qwen_block_and_rope_vs_diffusers():
    tq = pytest.importorskip('diffusers.models.transformers.transformer_qwenimage', reason='diffusers not installed: dit_ref.qwen_block stays unpinned')
    from oracle import dit_ref as D
    cfg = D.QwenCfg(num_layers=1, heads=2, joint_dim=128)
    w = {k: v.float() for k, v in D.make_qwen_weights(cfg, seed=6).items()}
    g = torch.Generator().manual_seed(7)
    hp, wp, T = 6, 4, 11
    img, txt, temb = torch.randn(1, hp * wp, cfg.dim, generator=g), torch.randn(1, T, cfg.dim, generator=g), torch.randn(1, cfg.dim, generator=g)
    ia, ta = D.qwen_rope_angles(hp, wp, T, cfg.axes_dims)
    ref_txt, ref_img = D.qwen_block(w, 'transformer_blocks.0.', cfg, img, txt, temb, (torch.cos(ia), torch.sin(ia)), (torch.cos(ta), torch.sin(ta)))
    blk = _load(tq.QwenImageTransformerBlock(dim=cfg.dim, num_attention_heads=cfg.heads, attention_head_dim=cfg.head_dim), w, 'transformer_blocks.0.')
    rope = tq.QwenEmbedRope(theta=10000, axes_dim=list(cfg.axes_dims), scale_rope=True)
    vid, tfreq = rope([(1, hp, wp)], [T], device=torch.device('cpu'))          # complex tables: angle = the restated angles
    assert _rel(torch.angle(vid).remainder(6.283185307179586), ia.remainder(6.283185307179586)) < 1e-4
    assert _rel(torch.angle(tfreq).remainder(6.283185307179586), ta.remainder(6.283185307179586)) < 1e-4
    with torch.no_grad():
        out_txt, out_img = blk(hidden_states=img, encoder_hidden_states=txt, encoder_hidden_states_mask=None, temb=temb,
                               image_rotary_emb=(vid, tfreq))
    assert _rel(ref_img, out_img) < TOL and _rel(ref_txt, out_txt) < TOL


# ------------------------------------------------------------------------------------------------------------- diffusers: VAE decoders
def test_flux_vae_decoder_vs_diffusers():
    diffusers = pytest.importorskip('diffusers', reason='diffusers not installed: oracle/vae_ref.py stays unpinned')
    from oracle import vae_ref as V
    chans = (32, 64, 64, 64)
    w = {k: v.float() for k, v in V.make_decoder_weights(chans, seed=1).items()}
    vae = diffusers.AutoencoderKL(in_channels=3, out_channels=3, latent_channels=16, block_out_channels=chans, layers_per_block=2,
                                  norm_num_groups=16, down_block_types=('DownEncoderBlock2D',) * 4, up_block_types=('UpDecoderBlock2D',) * 4,
                                  use_quant_conv=False, use_post_quant_conv=False, mid_block_add_attention=True)
    sd = {k: v for k, v in w.items() if k.startswith('decoder.')}
    missing, unexpected = vae.load_state_dict(sd, strict=False)
    assert not unexpected and all(not m.startswith('decoder.') for m in missing), (missing, unexpected)
    z = torch.randn(1, 16, 6, 5, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        out = vae.float().eval().decode(z, return_dict=False)[0]
    assert _rel(V.decode(w, z, chans, groups=16), out) < TOL


def test_qwen_vae_decoder_vs_diffusers():
    diffusers = pytest.importorskip('diffusers', reason='diffusers not installed: oracle/vae_qwen_ref.py stays unpinned')
    if not hasattr(diffusers, 'AutoencoderKLQwenImage'):
        pytest.skip('this diffusers has no AutoencoderKLQwenImage (needs >= 0.35)')
    from oracle import vae_qwen_ref as V
    w = {k: v.float() for k, v in V.make_decoder_weights(dim=32, seed=3).items()}
    vae = diffusers.AutoencoderKLQwenImage(base_dim=32, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                                           temperal_downsample=[False, True, True])
    missing, unexpected = vae.load_state_dict(w, strict=False)
    assert not unexpected and all(m.startswith(('encoder.', 'quant_conv.')) for m in missing), (missing, unexpected)
    z = torch.randn(1, 16, 1, 5, 4, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        out = vae.float().eval().decode(z, return_dict=False)[0][:, :, 0]
    assert _rel(V.decode(w, z[:, :, 0]), out) < TOL


# ------------------------------------------------------------------------------------------------------------- peft: the LoRA branch
def test_lora_linear_vs_peft():
    peft = pytest.importorskip('peft', reason='peft not installed: the LoRA branch of dit_ref.lin stays unpinned')
    from oracle import dit_ref as D
    torch.manual_seed(0)
    base = torch.nn.Sequential(torch.nn.Linear(48, 40))
    cfg = peft.LoraConfig(r=8, lora_alpha=8, lora_dropout=0.0, target_modules=['0'], init_lora_weights=False)    # alpha = r as arcflux.py:294-302
    m = peft.get_peft_model(base, cfg).eval()
    lay = m.base_model.model[0]
    A, B = lay.lora_A['default'].weight.detach(), lay.lora_B['default'].weight.detach()
    w = {'l.weight': lay.base_layer.weight.detach(), 'l.bias': lay.base_layer.bias.detach(), 'l.lora': (A, B, 1.0)}
    x = torch.randn(5, 48)
    with torch.no_grad():
        assert _rel(D.lin(w, 'l', x), m(x)) < 1e-5
    # merged form used by the inference engine: W + B A (scale alpha / r = 1)
    assert _rel(torch.nn.functional.linear(x, w['l.weight'] + B @ A, w['l.bias']), D.lin(w, 'l', x)) < 1e-5


# ------------------------------------------------------------------------------------------------------------- bitsandbytes: AdamW8bit
@pytest.mark.gpu
def test_adamw8bit_vs_bitsandbytes():
    bnb = pytest.importorskip('bitsandbytes', reason='bitsandbytes not installed: oracle/adamw8bit_ref.py stays unpinned')
    from arcflow_amd import ops
    from oracle import adamw8bit_ref as A8
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(8192, generator=g) * 0.1
    p_ref = torch.nn.Parameter(p0.clone().cuda())
    opt = bnb.optim.AdamW8bit([p_ref], lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0, min_8bit_size=4096)
    qmap1, qmap2 = ops.dynamic_map(True).cpu(), ops.dynamic_map(False).cpu()
    nb = p0.numel() // 256
    p = p0.clone()
    c1, c2 = torch.full((p.numel(),), int(qmap1.abs().argmin()), dtype=torch.uint8), torch.zeros(p.numel(), dtype=torch.uint8)
    a1, a2 = torch.zeros(nb), torch.zeros(nb)
    for step in range(1, 6):
        grad = torch.randn(p0.numel(), generator=g) * 0.01
        p_ref.grad = grad.clone().cuda()
        opt.step()
        p, c1, c2, a1, a2 = A8.adamw8bit_step(p, grad, c1, c2, a1, a2, qmap1, qmap2, 1e-3, step, (0.9, 0.95), 1e-8, 0.0)
    assert _rel(p, p_ref.detach().cpu()) < 1e-4
