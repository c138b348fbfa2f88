"""CPU oracle (torch fp32) for the denoiser forward: FLUX.1-dev MMDiT and Qwen-Image MMDiT
trunks with the three ArcFlow heads.

TEST INFRASTRUCTURE ONLY (see oracle/arcflow_ref.py header for the import rule).

What the reference owns and this file restates (paths under /root/reference/lakonlab):
  * models/architecture/arcflow/arcflux.py:134-257  embed -> 19 double -> 38 single -> norm_out
    -> proj_out_{means,logweights,loggamma}, log_softmax over K, timestep/guidance x1000,
    RoPE tables cast to the trunk dtype (:171-173)
  * models/architecture/arcflow/arcqwen.py:106-174  img_in / txt_norm+txt_in / 60 blocks / heads

What lives in an ABSENT third-party dependency: every block, norm, embedding and RoPE class is
imported from ``diffusers==0.35.1`` (reference requirements.txt:4; import sites arcflux.py:9-13,
arcqwen.py:9-11).  diffusers is not vendored under /root/reference and is not installable in the
build container, and the reference holds no test or golden vector for these modules.  The block
math below restates diffusers 0.35.1's published algorithm (FluxTransformerBlock,
FluxSingleTransformerBlock, QwenImageTransformerBlock, AdaLayerNormZero/-Single/-Continuous,
FluxPosEmbed, QwenEmbedRope(scale_rope=True), Timesteps/TimestepEmbedding, FeedForward
gelu-approximate, RMSNorm) as summarised in SURVEY.md Appendix B.

Parity status of THIS file: **parity unpinned** for the diffusers-owned block math (no reference
vector exists to pin it to); the ArcFlow-owned parts (three heads, reshape to [B,N,K,*],
log_softmax over K, x1000 scaling) follow the reference source directly.

Weights are passed as a plain ``dict[str, Tensor]`` with the diffusers state-dict key names, so
that the same dict drives this oracle and the HIP engine in the parity tests.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LN_EPS = 1e-6


@dataclass
class FluxCfg:
    num_layers: int = 19
    num_single_layers: int = 38
    heads: int = 24
    head_dim: int = 128
    in_channels: int = 64
    joint_dim: int = 4096
    pooled_dim: int = 768
    num_gaussians: int = 16
    logweights_channels: int = 4
    axes_dims: Tuple[int, int, int] = (16, 56, 56)
    guidance_embeds: bool = True
    mlp_ratio: int = 4

    @property
    def dim(self):
        return self.heads * self.head_dim


@dataclass
class QwenCfg:
    num_layers: int = 60
    heads: int = 24
    head_dim: int = 128
    in_channels: int = 64
    joint_dim: int = 3584
    num_gaussians: int = 16
    logweights_channels: int = 4
    axes_dims: Tuple[int, int, int] = (16, 56, 56)
    mlp_ratio: int = 4

    @property
    def dim(self):
        return self.heads * self.head_dim


# ----------------------------------------------------------------------------- primitives
def lin(w: Dict[str, Tensor], name: str, x: Tensor) -> Tensor:
    """nn.Linear; with ``w[name + '.lora'] = (A [r,in], B [out,r], keep_scale)`` also the peft LoRA branch with input dropout
    (peft 0.17 LoraLayer.forward, configured at lakonlab/models/architecture/arcflow/arcflux.py:294-302, alpha = r):
    y = W x + b + B A (x * keep_scale), keep_scale = keep / (1 - p) broadcastable to x (the dropout draw is an input)."""
    b = w.get(name + '.bias')
    y = F.linear(x, w[name + '.weight'].float(), None if b is None else b.float())
    if name + '.lora' in w:
        a_, b_, keep_scale = w[name + '.lora']
        y = y + F.linear(F.linear(x * keep_scale, a_), b_)
    return y


def layer_norm(x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), eps=LN_EPS)


def rms_norm(x: Tensor, weight: Optional[Tensor], eps: float = 1e-6) -> Tensor:
    y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return y if weight is None else y * weight.float()


def gelu_tanh(x: Tensor) -> Tensor:
    return F.gelu(x, approximate='tanh')


def sincos_embedding(t: Tensor, dim: int = 256, scale: float = 1.0, max_period: float = 10000.0) -> Tensor:
    """diffusers Timesteps(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    ang = scale * (t.float()[:, None] * freqs[None, :])
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def mlp_embed(w, prefix: str, x: Tensor) -> Tensor:
    """linear_1 -> SiLU -> linear_2 (TimestepEmbedding / PixArtAlphaTextProjection)."""
    return lin(w, prefix + '.linear_2', F.silu(lin(w, prefix + '.linear_1', x)))


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """Interleaved-pair rotation: x [B,S,H,D], cos/sin [S,D/2].
    out[2i] = x[2i] c_i - x[2i+1] s_i ; out[2i+1] = x[2i] s_i + x[2i+1] c_i."""
    xe, xo = x[..., 0::2], x[..., 1::2]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    out = torch.stack([xe * c - xo * s, xe * s + xo * c], dim=-1)
    return out.flatten(-2)


def attention(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """q,k,v [B,S,H,D] -> [B,S,H*D]; softmax(q k^T / sqrt(D)) v, no mask."""
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    return o.transpose(1, 2).flatten(2)


# ----------------------------------------------------------------------------- RoPE tables
def flux_rope_angles(ids: Tensor, axes_dims: Sequence[int], theta: float = 10000.0) -> Tensor:
    """FluxPosEmbed angles [S, sum(axes)/2] in fp64: per axis a, pos_a * theta^(-2i/d_a)."""
    out = []
    for a, d in enumerate(axes_dims):
        omega = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        out.append(ids[:, a].double()[:, None] * omega[None, :])
    return torch.cat(out, dim=-1)


def flux_ids(hp: int, wp: int, txt_len: int) -> Tensor:
    """cat(txt_ids, img_ids): text (0,0,0); image (0,row,col)  (arcflux.py:360-373,426-428)."""
    img = torch.zeros(hp, wp, 3)
    img[..., 1] += torch.arange(hp)[:, None]
    img[..., 2] += torch.arange(wp)[None, :]
    return torch.cat([torch.zeros(txt_len, 3), img.reshape(hp * wp, 3)], dim=0)


def flux_rope_tables(hp: int, wp: int, txt_len: int, axes_dims=(16, 56, 56), bf16_round: bool = True
                     ) -> Tuple[Tensor, Tensor]:
    ang = flux_rope_angles(flux_ids(hp, wp, txt_len), axes_dims)
    cos, sin = torch.cos(ang).float(), torch.sin(ang).float()
    if bf16_round:      # arcflux.py:173 casts the tables to the trunk dtype (bf16)
        cos, sin = cos.bfloat16().float(), sin.bfloat16().float()
    return cos, sin


def qwen_rope_angles(hp: int, wp: int, txt_len: int, axes_dims=(16, 56, 56), theta: float = 10000.0
                     ) -> Tuple[Tensor, Tensor]:
    """QwenEmbedRope(scale_rope=True), one frame: image positions are centred
    (rows -(h-h//2)..h//2-1), text positions start at max(h//2, w//2) on all three axes.
    Returns (img_angles [hp*wp, 64], txt_angles [T, 64]) in fp32 like the reference's tables."""
    def omega(d):
        return 1.0 / torch.pow(torch.tensor(theta), torch.arange(0, d, 2, dtype=torch.float32) / d)
    om = [omega(d) for d in axes_dims]

    def centred(n):
        return torch.cat([torch.arange(-(n - n // 2), 0), torch.arange(0, n // 2)]).float()
    fr = torch.zeros(hp, wp, 1) * om[0]                                  # frame index 0
    fh = (centred(hp)[:, None] * om[1][None, :])[:, None, :].expand(hp, wp, -1)
    fw = (centred(wp)[:, None] * om[2][None, :])[None, :, :].expand(hp, wp, -1)
    img = torch.cat([fr.expand(hp, wp, -1), fh, fw], dim=-1).reshape(hp * wp, -1)
    start = max(hp // 2, wp // 2)
    pos = torch.arange(start, start + txt_len).float()
    txt = torch.cat([pos[:, None] * o[None, :] for o in om], dim=-1)
    return img, txt


# ----------------------------------------------------------------------------- FLUX
def flux_temb(w, cfg: FluxCfg, timestep: Tensor, guidance: Optional[Tensor], pooled: Tensor) -> Tensor:
    t = mlp_embed(w, 'time_text_embed.timestep_embedder', sincos_embedding(timestep * 1000))
    if cfg.guidance_embeds:
        t = t + mlp_embed(w, 'time_text_embed.guidance_embedder', sincos_embedding(guidance * 1000))
    return t + mlp_embed(w, 'time_text_embed.text_embedder', pooled.float())


def _heads(x: Tensor, h: int) -> Tensor:
    return x.unflatten(-1, (h, -1))


def flux_double_block(w, p: str, cfg: FluxCfg, img: Tensor, txt: Tensor, temb: Tensor,
                      cos: Tensor, sin: Tensor) -> Tuple[Tensor, Tensor]:
    """FluxTransformerBlock: returns (txt, img)."""
    h = cfg.heads
    e = F.silu(temb)
    i_sh1, i_sc1, i_g1, i_sh2, i_sc2, i_g2 = lin(w, p + 'norm1.linear', e).chunk(6, dim=1)
    t_sh1, t_sc1, t_g1, t_sh2, t_sc2, t_g2 = lin(w, p + 'norm1_context.linear', e).chunk(6, dim=1)
    xi = layer_norm(img) * (1 + i_sc1[:, None]) + i_sh1[:, None]
    xt = layer_norm(txt) * (1 + t_sc1[:, None]) + t_sh1[:, None]
    q = rms_norm(_heads(lin(w, p + 'attn.to_q', xi), h), w[p + 'attn.norm_q.weight'])
    k = rms_norm(_heads(lin(w, p + 'attn.to_k', xi), h), w[p + 'attn.norm_k.weight'])
    v = _heads(lin(w, p + 'attn.to_v', xi), h)
    qt = rms_norm(_heads(lin(w, p + 'attn.add_q_proj', xt), h), w[p + 'attn.norm_added_q.weight'])
    kt = rms_norm(_heads(lin(w, p + 'attn.add_k_proj', xt), h), w[p + 'attn.norm_added_k.weight'])
    vt = _heads(lin(w, p + 'attn.add_v_proj', xt), h)
    q = apply_rope(torch.cat([qt, q], dim=1), cos, sin)
    k = apply_rope(torch.cat([kt, k], dim=1), cos, sin)
    o = attention(q, k, torch.cat([vt, v], dim=1))
    T = txt.shape[1]
    ot, oi = o[:, :T], o[:, T:]
    img = img + i_g1[:, None] * lin(w, p + 'attn.to_out.0', oi)
    txt = txt + t_g1[:, None] * lin(w, p + 'attn.to_add_out', ot)
    xi = layer_norm(img) * (1 + i_sc2[:, None]) + i_sh2[:, None]
    img = img + i_g2[:, None] * lin(w, p + 'ff.net.2', gelu_tanh(lin(w, p + 'ff.net.0.proj', xi)))
    xt = layer_norm(txt) * (1 + t_sc2[:, None]) + t_sh2[:, None]
    txt = txt + t_g2[:, None] * lin(w, p + 'ff_context.net.2', gelu_tanh(lin(w, p + 'ff_context.net.0.proj', xt)))
    return txt, img


def flux_single_block(w, p: str, cfg: FluxCfg, x: Tensor, temb: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """FluxSingleTransformerBlock on the joint [text; image] sequence."""
    h = cfg.heads
    sh, sc, g = lin(w, p + 'norm.linear', F.silu(temb)).chunk(3, dim=1)
    xn = layer_norm(x) * (1 + sc[:, None]) + sh[:, None]
    mlp = gelu_tanh(lin(w, p + 'proj_mlp', xn))
    q = apply_rope(rms_norm(_heads(lin(w, p + 'attn.to_q', xn), h), w[p + 'attn.norm_q.weight']), cos, sin)
    k = apply_rope(rms_norm(_heads(lin(w, p + 'attn.to_k', xn), h), w[p + 'attn.norm_k.weight']), cos, sin)
    v = _heads(lin(w, p + 'attn.to_v', xn), h)
    o = attention(q, k, v)
    return x + g[:, None] * lin(w, p + 'proj_out', torch.cat([o, mlp], dim=2))


def arc_heads(w, temb: Tensor, x: Tensor, K: int, out_ch: int, lw_ch: int):
    """norm_out (AdaLayerNormContinuous, scale first) + three heads + log_softmax over K
    (arcflux.py:241-257)."""
    sc, sh = lin(w, 'norm_out.linear', F.silu(temb)).chunk(2, dim=1)
    x = layer_norm(x) * (1 + sc[:, None]) + sh[:, None]
    b, n, _ = x.shape
    means = lin(w, 'proj_out_means', x).reshape(b, n, K, out_ch)
    logw = lin(w, 'proj_out_logweights', x).reshape(b, n, K, lw_ch).log_softmax(dim=-2)
    logg = lin(w, 'proj_out_loggamma', x).reshape(b, n, K - 1, lw_ch)
    return means, logw, logg


def flux_forward(w: Dict[str, Tensor], cfg: FluxCfg, hidden: Tensor, ctx: Tensor, pooled: Tensor,
                 timestep: Tensor, guidance: Optional[Tensor], hp: int, wp: int,
                 rope_bf16: bool = True, return_trunk: bool = False):
    """_ArcFluxTransformer2DModel.forward (arcflux.py:134-257).
    hidden [B,N,64], ctx [B,T,joint], pooled [B,768], timestep = sigma in [0,1], guidance e.g. 3.5."""
    img = lin(w, 'x_embedder', hidden.float())
    temb = flux_temb(w, cfg, timestep, guidance, pooled)
    txt = lin(w, 'context_embedder', ctx.float())
    T = txt.shape[1]
    cos, sin = flux_rope_tables(hp, wp, T, cfg.axes_dims, rope_bf16)
    for i in range(cfg.num_layers):
        txt, img = flux_double_block(w, f'transformer_blocks.{i}.', cfg, img, txt, temb, cos, sin)
    x = torch.cat([txt, img], dim=1)
    for i in range(cfg.num_single_layers):
        x = flux_single_block(w, f'single_transformer_blocks.{i}.', cfg, x, temb, cos, sin)
    img = x[:, T:]
    if return_trunk:
        return img
    return arc_heads(w, temb, img, cfg.num_gaussians, cfg.in_channels, cfg.logweights_channels)


def flux_teacher_forward(w, cfg: FluxCfg, hidden, ctx, pooled, timestep, guidance, hp, wp, rope_bf16=True):
    """Plain FLUX (teacher) forward: same trunk, single proj_out head (diffusers/flux.py:122-156)."""
    img = flux_forward(w, cfg, hidden, ctx, pooled, timestep, guidance, hp, wp, rope_bf16, return_trunk=True)
    temb = flux_temb(w, cfg, timestep, guidance, pooled)
    sc, sh = lin(w, 'norm_out.linear', F.silu(temb)).chunk(2, dim=1)
    return lin(w, 'proj_out', layer_norm(img) * (1 + sc[:, None]) + sh[:, None])


# ----------------------------------------------------------------------------- Qwen-Image
def qwen_block(w, p: str, cfg: QwenCfg, img: Tensor, txt: Tensor, temb: Tensor,
               rope_img: Tuple[Tensor, Tensor], rope_txt: Tuple[Tensor, Tensor]) -> Tuple[Tensor, Tensor]:
    """QwenImageTransformerBlock: returns (txt, img)."""
    h = cfg.heads
    e = F.silu(temb)
    im1, im2 = lin(w, p + 'img_mod.1', e).chunk(2, dim=-1)
    tm1, tm2 = lin(w, p + 'txt_mod.1', e).chunk(2, dim=-1)

    def modulate(x, mod):
        sh, sc, g = mod.chunk(3, dim=-1)
        return layer_norm(x) * (1 + sc[:, None]) + sh[:, None], g[:, None]
    xi, gi1 = modulate(img, im1)
    xt, gt1 = modulate(txt, tm1)
    q = apply_rope(rms_norm(_heads(lin(w, p + 'attn.to_q', xi), h), w[p + 'attn.norm_q.weight']), *rope_img)
    k = apply_rope(rms_norm(_heads(lin(w, p + 'attn.to_k', xi), h), w[p + 'attn.norm_k.weight']), *rope_img)
    v = _heads(lin(w, p + 'attn.to_v', xi), h)
    qt = apply_rope(rms_norm(_heads(lin(w, p + 'attn.add_q_proj', xt), h), w[p + 'attn.norm_added_q.weight']), *rope_txt)
    kt = apply_rope(rms_norm(_heads(lin(w, p + 'attn.add_k_proj', xt), h), w[p + 'attn.norm_added_k.weight']), *rope_txt)
    vt = _heads(lin(w, p + 'attn.add_v_proj', xt), h)
    o = attention(torch.cat([qt, q], 1), torch.cat([kt, k], 1), torch.cat([vt, v], 1))
    T = txt.shape[1]
    img = img + gi1 * lin(w, p + 'attn.to_out.0', o[:, T:])
    txt = txt + gt1 * lin(w, p + 'attn.to_add_out', o[:, :T])
    xi, gi2 = modulate(img, im2)
    img = img + gi2 * lin(w, p + 'img_mlp.net.2', gelu_tanh(lin(w, p + 'img_mlp.net.0.proj', xi)))
    xt, gt2 = modulate(txt, tm2)
    txt = txt + gt2 * lin(w, p + 'txt_mlp.net.2', gelu_tanh(lin(w, p + 'txt_mlp.net.0.proj', xt)))
    return txt, img


def qwen_forward(w: Dict[str, Tensor], cfg: QwenCfg, hidden: Tensor, ctx: Tensor, timestep: Tensor,
                 hp: int, wp: int):
    """_ArcQwenImageTransformer2DModel.forward (arcqwen.py:106-174).  ctx [B,T,3584] holds only the
    real (unpadded) text tokens; timestep = sigma (x1000 inside the sinusoid, arcqwen.py:128)."""
    img = lin(w, 'img_in', hidden.float())
    txt = lin(w, 'txt_in', rms_norm(ctx.float(), w['txt_norm.weight']))
    temb = mlp_embed(w, 'time_text_embed.timestep_embedder', sincos_embedding(timestep, scale=1000.0))
    ia, ta = qwen_rope_angles(hp, wp, txt.shape[1], cfg.axes_dims)
    rope_img, rope_txt = (torch.cos(ia), torch.sin(ia)), (torch.cos(ta), torch.sin(ta))
    for i in range(cfg.num_layers):
        txt, img = qwen_block(w, f'transformer_blocks.{i}.', cfg, img, txt, temb, rope_img, rope_txt)
    return arc_heads(w, temb, img, cfg.num_gaussians, cfg.in_channels, cfg.logweights_channels)


# ----------------------------------------------------------------------------- synthetic weights
def _lin_init(w, name, out_f, in_f, gen, std=0.02, bias_std=0.02, dtype=torch.bfloat16):
    w[name + '.weight'] = (torch.randn(out_f, in_f, generator=gen) * std).to(dtype)
    w[name + '.bias'] = (torch.randn(out_f, generator=gen) * bias_std).to(dtype)


def make_flux_weights(cfg: FluxCfg, seed: int = 0, dtype=torch.bfloat16, teacher_head: bool = False
                      ) -> Dict[str, Tensor]:
    """Random FLUX-architecture weights with the diffusers key names (N(0,0.02^2); RMSNorm weights
    1+N(0,0.02^2); modulation / gate linears N(0,0.02^2) with a larger bias so gates are not ~0)."""
    g = torch.Generator().manual_seed(seed)
    D, w = cfg.dim, {}
    _lin_init(w, 'x_embedder', D, cfg.in_channels, g, std=0.1, dtype=dtype)
    _lin_init(w, 'context_embedder', D, cfg.joint_dim, g, dtype=dtype)
    names = ['timestep_embedder', 'text_embedder'] + (['guidance_embedder'] if cfg.guidance_embeds else [])
    for nm in names:
        in1 = cfg.pooled_dim if nm == 'text_embedder' else 256
        _lin_init(w, f'time_text_embed.{nm}.linear_1', D, in1, g, std=0.05, dtype=dtype)
        _lin_init(w, f'time_text_embed.{nm}.linear_2', D, D, g, std=0.03, dtype=dtype)

    def rmsw(name):
        w[name] = (1 + 0.02 * torch.randn(cfg.head_dim, generator=g)).to(dtype)
    for i in range(cfg.num_layers):
        p = f'transformer_blocks.{i}.'
        _lin_init(w, p + 'norm1.linear', 6 * D, D, g, bias_std=0.3, dtype=dtype)
        _lin_init(w, p + 'norm1_context.linear', 6 * D, D, g, bias_std=0.3, dtype=dtype)
        for nm in ('to_q', 'to_k', 'to_v', 'add_q_proj', 'add_k_proj', 'add_v_proj', 'to_out.0', 'to_add_out'):
            _lin_init(w, p + 'attn.' + nm, D, D, g, dtype=dtype)
        for nm in ('norm_q', 'norm_k', 'norm_added_q', 'norm_added_k'):
            rmsw(p + f'attn.{nm}.weight')
        for ff in ('ff', 'ff_context'):
            _lin_init(w, p + ff + '.net.0.proj', cfg.mlp_ratio * D, D, g, dtype=dtype)
            _lin_init(w, p + ff + '.net.2', D, cfg.mlp_ratio * D, g, dtype=dtype)
    for i in range(cfg.num_single_layers):
        p = f'single_transformer_blocks.{i}.'
        _lin_init(w, p + 'norm.linear', 3 * D, D, g, bias_std=0.3, dtype=dtype)
        for nm in ('to_q', 'to_k', 'to_v'):
            _lin_init(w, p + 'attn.' + nm, D, D, g, dtype=dtype)
        rmsw(p + 'attn.norm_q.weight')
        rmsw(p + 'attn.norm_k.weight')
        _lin_init(w, p + 'proj_mlp', cfg.mlp_ratio * D, D, g, dtype=dtype)
        _lin_init(w, p + 'proj_out', D, (1 + cfg.mlp_ratio) * D, g, dtype=dtype)
    _lin_init(w, 'norm_out.linear', 2 * D, D, g, bias_std=0.1, dtype=dtype)
    K, C, L = cfg.num_gaussians, cfg.in_channels, cfg.logweights_channels
    _lin_init(w, 'proj_out_means', K * C, D, g, dtype=dtype)
    _lin_init(w, 'proj_out_logweights', K * L, D, g, dtype=dtype)
    _lin_init(w, 'proj_out_loggamma', (K - 1) * L, D, g, bias_std=0.5, dtype=dtype)
    if teacher_head:
        _lin_init(w, 'proj_out', C, D, g, dtype=dtype)
    return w


def make_qwen_weights(cfg: QwenCfg, seed: int = 0, dtype=torch.bfloat16) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    D, w = cfg.dim, {}
    _lin_init(w, 'img_in', D, cfg.in_channels, g, std=0.1, dtype=dtype)
    _lin_init(w, 'txt_in', D, cfg.joint_dim, g, dtype=dtype)
    w['txt_norm.weight'] = (1 + 0.02 * torch.randn(cfg.joint_dim, generator=g)).to(dtype)
    _lin_init(w, 'time_text_embed.timestep_embedder.linear_1', D, 256, g, std=0.05, dtype=dtype)
    _lin_init(w, 'time_text_embed.timestep_embedder.linear_2', D, D, g, std=0.03, dtype=dtype)
    for i in range(cfg.num_layers):
        p = f'transformer_blocks.{i}.'
        _lin_init(w, p + 'img_mod.1', 6 * D, D, g, bias_std=0.3, dtype=dtype)
        _lin_init(w, p + 'txt_mod.1', 6 * D, D, g, bias_std=0.3, dtype=dtype)
        for nm in ('to_q', 'to_k', 'to_v', 'add_q_proj', 'add_k_proj', 'add_v_proj', 'to_out.0', 'to_add_out'):
            _lin_init(w, p + 'attn.' + nm, D, D, g, dtype=dtype)
        for nm in ('norm_q', 'norm_k', 'norm_added_q', 'norm_added_k'):
            w[p + f'attn.{nm}.weight'] = (1 + 0.02 * torch.randn(cfg.head_dim, generator=g)).to(dtype)
        for ff in ('img_mlp', 'txt_mlp'):
            _lin_init(w, p + ff + '.net.0.proj', cfg.mlp_ratio * D, D, g, dtype=dtype)
            _lin_init(w, p + ff + '.net.2', D, cfg.mlp_ratio * D, g, dtype=dtype)
    _lin_init(w, 'norm_out.linear', 2 * D, D, g, bias_std=0.1, dtype=dtype)
    K, C, L = cfg.num_gaussians, cfg.in_channels, cfg.logweights_channels
    _lin_init(w, 'proj_out_means', K * C, D, g, dtype=dtype)
    _lin_init(w, 'proj_out_logweights', K * L, D, g, dtype=dtype)
    _lin_init(w, 'proj_out_loggamma', (K - 1) * L, D, g, bias_std=0.5, dtype=dtype)
    return w
