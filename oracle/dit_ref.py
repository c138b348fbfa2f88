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

Two evaluation modes.  Default: fp32 throughout (the mathematical reference).  ``with eager_bf16():``
the same op sequence with EVERY op output rounded to bf16 -- how the reference actually runs
(``torch_dtype=torch.bfloat16`` modules in eager mode, inference_flux.py:6-8: each nn.Linear / LayerNorm /
elementwise op / SDPA returns a bf16 tensor; norms, softmax and the rotary product compute in fp32
internally and round once; SURVEY App. B "Rounding").  The full-depth parity tests measure the HIP
engine's distance to the fp32 evaluation against THIS mode's distance to it.  All functions run on the
device of their inputs (the full-size tests evaluate the oracle on the GPU: test infrastructure, never
the product path).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LN_EPS = 1e-6
_EAGER_BF16 = False


class eager_bf16:
    """Context manager: every op output of this module is rounded to bf16 (values stay in fp32 storage)."""

    def __enter__(self):
        global _EAGER_BF16
        self._prev, _EAGER_BF16 = _EAGER_BF16, True
        return self

    def __exit__(self, *exc):
        global _EAGER_BF16
        _EAGER_BF16 = self._prev


def _r(x: Tensor) -> Tensor:
    return x.bfloat16().float() if _EAGER_BF16 else x


MODEL_DTYPE = torch.bfloat16


def cond_cast(x: Tensor) -> Tensor:
    """The reference's EXPLICIT casts of the conditioning scalars to the trunk dtype, part of the algorithm in both modes:
    ``timestep.to(hidden_states.dtype) * 1000`` / ``guidance.to(hidden_states.dtype) * 1000`` (arcflux.py:160-162; the product is a
    bf16 tensor too) and ``timestep.to(hidden_states.dtype)`` (arcqwen.py:128), with the released pipelines' bf16 transformers
    (inference_flux.py:6-8) and under the training autocast (arcflux.py:437-440).  sigma = 0.76190 therefore reaches FLUX's
    sinusoid as 760.0 (bf16 spacing 4 in [512, 1024)) and guidance 3.5 as 3504: at the top frequencies that is another
    embedding, so an implementation that feeds 761.9 / 3500 is NOT evaluating the network the reference evaluates."""
    return x.float().to(MODEL_DTYPE).float()


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
    y = _r(F.linear(x, w[name + '.weight'].float(), None if b is None else b.float()))
    if name + '.lora' in w:
        a_, b_, keep_scale = w[name + '.lora']
        y = _r(y + _r(F.linear(_r(F.linear(_r(x * keep_scale), a_)), b_)))
    return y


def layer_norm(x: Tensor) -> Tensor:
    return _r(F.layer_norm(x, (x.shape[-1],), eps=LN_EPS))


def modulate(x: Tensor, scale: Tensor, shift: Tensor) -> Tensor:
    """LN(x) * (1 + scale) + shift with [B, D] vectors; three elementwise ops in the eager module."""
    return _r(_r(layer_norm(x) * _r(1 + scale[:, None])) + shift[:, None])


def gated_add(x: Tensor, gate: Tensor, y: Tensor) -> Tensor:
    """x + gate * y  (two elementwise ops)."""
    return _r(x + _r(gate[:, None] * y))


def rms_norm(x: Tensor, weight: Optional[Tensor], eps: float = 1e-6) -> Tensor:
    y = _r(x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))       # diffusers RMSNorm: fp32 inside, cast, then * weight
    return y if weight is None else _r(y * weight.float())


def gelu_tanh(x: Tensor) -> Tensor:
    return _r(F.gelu(x, approximate='tanh'))


def sincos_embedding(t: Tensor, dim: int = 256, scale: float = 1.0, max_period: float = 10000.0) -> Tensor:
    """diffusers Timesteps(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(t.device)
    ang = scale * (t.float()[:, None] * freqs[None, :])
    return _r(torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1))


def mlp_embed(w, prefix: str, x: Tensor) -> Tensor:
    """linear_1 -> SiLU -> linear_2 (TimestepEmbedding / PixArtAlphaTextProjection)."""
    return lin(w, prefix + '.linear_2', _r(F.silu(lin(w, prefix + '.linear_1', x))))


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """Interleaved-pair rotation: x [B,S,H,D], cos/sin [S,D/2].
    out[2i] = x[2i] c_i - x[2i+1] s_i ; out[2i+1] = x[2i] s_i + x[2i+1] c_i."""
    xe, xo = x[..., 0::2], x[..., 1::2]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    out = torch.stack([xe * c - xo * s, xe * s + xo * c], dim=-1)       # fp32 product, one rounding (apply_rotary_emb)
    return _r(out.flatten(-2))


def attention(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """q,k,v [B,S,H,D] -> [B,S,H*D]; softmax(q k^T / sqrt(D)) v, no mask.  On the CPU in fp32 mode this is torch's SDPA; on a
    device (or in eager-bf16 mode) the same product head group by head group, in eager-bf16 mode with the probabilities
    rounded to bf16 in front of P V as the fused bf16 SDPA kernels do."""
    qt, kt, vt = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    if not (_EAGER_BF16 or q.is_cuda):
        o = F.scaled_dot_product_attention(qt, kt, vt)
        return o.transpose(1, 2).flatten(2)
    o = torch.empty_like(qt)
    hs = max(1, (1 << 27) // (q.shape[1] * k.shape[1]))             # <= 512 MB of fp32 scores at a time
    for h0 in range(0, qt.shape[1], hs):
        p = torch.softmax(torch.matmul(qt[:, h0:h0 + hs], kt[:, h0:h0 + hs].transpose(-1, -2)) * (q.shape[-1] ** -0.5), dim=-1)
        o[:, h0:h0 + hs] = torch.matmul(_r(p), vt[:, h0:h0 + hs])
    return _r(o.transpose(1, 2).flatten(2))


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
    """arcflux.py:160-168 (timestep / guidance cast to the trunk dtype and scaled there: ``cond_cast``)."""
    t = mlp_embed(w, 'time_text_embed.timestep_embedder', sincos_embedding(cond_cast(cond_cast(timestep) * 1000)))
    if cfg.guidance_embeds:
        t = _r(t + mlp_embed(w, 'time_text_embed.guidance_embedder', sincos_embedding(cond_cast(cond_cast(guidance) * 1000))))
    return _r(t + mlp_embed(w, 'time_text_embed.text_embedder', pooled.float()))


def _heads(x: Tensor, h: int) -> Tensor:
    return x.unflatten(-1, (h, -1))


def flux_double_block(w, p: str, cfg: FluxCfg, img: Tensor, txt: Tensor, temb: Tensor,
                      cos: Tensor, sin: Tensor) -> Tuple[Tensor, Tensor]:
    """FluxTransformerBlock: returns (txt, img)."""
    h = cfg.heads
    e = _r(F.silu(temb))
    i_sh1, i_sc1, i_g1, i_sh2, i_sc2, i_g2 = lin(w, p + 'norm1.linear', e).chunk(6, dim=1)
    t_sh1, t_sc1, t_g1, t_sh2, t_sc2, t_g2 = lin(w, p + 'norm1_context.linear', e).chunk(6, dim=1)
    xi = modulate(img, i_sc1, i_sh1)
    xt = modulate(txt, t_sc1, t_sh1)
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
    img = gated_add(img, i_g1, lin(w, p + 'attn.to_out.0', oi))
    txt = gated_add(txt, t_g1, lin(w, p + 'attn.to_add_out', ot))
    xi = modulate(img, i_sc2, i_sh2)
    img = gated_add(img, i_g2, lin(w, p + 'ff.net.2', gelu_tanh(lin(w, p + 'ff.net.0.proj', xi))))
    xt = modulate(txt, t_sc2, t_sh2)
    txt = gated_add(txt, t_g2, lin(w, p + 'ff_context.net.2', gelu_tanh(lin(w, p + 'ff_context.net.0.proj', xt))))
    return txt, img


def flux_single_block(w, p: str, cfg: FluxCfg, x: Tensor, temb: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """FluxSingleTransformerBlock on the joint [text; image] sequence."""
    h = cfg.heads
    sh, sc, g = lin(w, p + 'norm.linear', _r(F.silu(temb))).chunk(3, dim=1)
    xn = modulate(x, sc, sh)
    mlp = gelu_tanh(lin(w, p + 'proj_mlp', xn))
    q = apply_rope(rms_norm(_heads(lin(w, p + 'attn.to_q', xn), h), w[p + 'attn.norm_q.weight']), cos, sin)
    k = apply_rope(rms_norm(_heads(lin(w, p + 'attn.to_k', xn), h), w[p + 'attn.norm_k.weight']), cos, sin)
    v = _heads(lin(w, p + 'attn.to_v', xn), h)
    o = attention(q, k, v)
    return gated_add(x, g, lin(w, p + 'proj_out', torch.cat([o, mlp], dim=2)))


LAST_NORM_OUT_INPUT = None


def arc_heads(w, temb: Tensor, x: Tensor, K: int, out_ch: int, lw_ch: int):
    """norm_out (AdaLayerNormContinuous, scale first) + three heads + log_softmax over K
    (arcflux.py:241-257)."""
    global LAST_NORM_OUT_INPUT
    LAST_NORM_OUT_INPUT = x.detach()          # (tests report how heavy-tailed the residual stream entering norm_out is)
    sc, sh = lin(w, 'norm_out.linear', _r(F.silu(temb))).chunk(2, dim=1)
    x = modulate(x, sc, sh)
    b, n, _ = x.shape
    means = lin(w, 'proj_out_means', x).reshape(b, n, K, out_ch)
    logw = _r(lin(w, 'proj_out_logweights', x).reshape(b, n, K, lw_ch).log_softmax(dim=-2))
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
    cos, sin = (t.to(img.device) for t in flux_rope_tables(hp, wp, T, cfg.axes_dims, rope_bf16))
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
    sc, sh = lin(w, 'norm_out.linear', _r(F.silu(temb))).chunk(2, dim=1)
    return lin(w, 'proj_out', modulate(img, sc, sh))


# ----------------------------------------------------------------------------- Qwen-Image
def qwen_block(w, p: str, cfg: QwenCfg, img: Tensor, txt: Tensor, temb: Tensor,
               rope_img: Tuple[Tensor, Tensor], rope_txt: Tuple[Tensor, Tensor]) -> Tuple[Tensor, Tensor]:
    """QwenImageTransformerBlock: returns (txt, img)."""
    h = cfg.heads
    e = _r(F.silu(temb))
    im1, im2 = lin(w, p + 'img_mod.1', e).chunk(2, dim=-1)
    tm1, tm2 = lin(w, p + 'txt_mod.1', e).chunk(2, dim=-1)

    def mod3(x, mod):
        sh, sc, g = mod.chunk(3, dim=-1)
        return modulate(x, sc, sh), g
    xi, gi1 = mod3(img, im1)
    xt, gt1 = mod3(txt, tm1)
    q = apply_rope(rms_norm(_heads(lin(w, p + 'attn.to_q', xi), h), w[p + 'attn.norm_q.weight']), *rope_img)
    k = apply_rope(rms_norm(_heads(lin(w, p + 'attn.to_k', xi), h), w[p + 'attn.norm_k.weight']), *rope_img)
    v = _heads(lin(w, p + 'attn.to_v', xi), h)
    qt = apply_rope(rms_norm(_heads(lin(w, p + 'attn.add_q_proj', xt), h), w[p + 'attn.norm_added_q.weight']), *rope_txt)
    kt = apply_rope(rms_norm(_heads(lin(w, p + 'attn.add_k_proj', xt), h), w[p + 'attn.norm_added_k.weight']), *rope_txt)
    vt = _heads(lin(w, p + 'attn.add_v_proj', xt), h)
    o = attention(torch.cat([qt, q], 1), torch.cat([kt, k], 1), torch.cat([vt, v], 1))
    T = txt.shape[1]
    img = gated_add(img, gi1, lin(w, p + 'attn.to_out.0', o[:, T:]))
    txt = gated_add(txt, gt1, lin(w, p + 'attn.to_add_out', o[:, :T]))
    xi, gi2 = mod3(img, im2)
    img = gated_add(img, gi2, lin(w, p + 'img_mlp.net.2', gelu_tanh(lin(w, p + 'img_mlp.net.0.proj', xi))))
    xt, gt2 = mod3(txt, tm2)
    txt = gated_add(txt, gt2, lin(w, p + 'txt_mlp.net.2', gelu_tanh(lin(w, p + 'txt_mlp.net.0.proj', xt))))
    return txt, img


def qwen_forward(w: Dict[str, Tensor], cfg: QwenCfg, hidden: Tensor, ctx: Tensor, timestep: Tensor,
                 hp: int, wp: int):
    """_ArcQwenImageTransformer2DModel.forward (arcqwen.py:106-174).  ctx [B,T,3584] holds only the
    real (unpadded) text tokens; timestep = sigma (x1000 inside the sinusoid, arcqwen.py:128)."""
    img = lin(w, 'img_in', hidden.float())
    txt = lin(w, 'txt_in', rms_norm(ctx.float(), w['txt_norm.weight']))
    temb = mlp_embed(w, 'time_text_embed.timestep_embedder', sincos_embedding(cond_cast(timestep), scale=1000.0))
    ia, ta = (t.to(img.device) for t in qwen_rope_angles(hp, wp, txt.shape[1], cfg.axes_dims))
    rope_img, rope_txt = (torch.cos(ia), torch.sin(ia)), (torch.cos(ta), torch.sin(ta))
    for i in range(cfg.num_layers):
        txt, img = qwen_block(w, f'transformer_blocks.{i}.', cfg, img, txt, temb, rope_img, rope_txt)
    return arc_heads(w, temb, img, cfg.num_gaussians, cfg.in_channels, cfg.logweights_channels)


# ----------------------------------------------------------------------------- synthetic weights
def _lin_init(w, name, out_f, in_f, gen, std=0.02, bias_std=0.02, dtype=torch.bfloat16):
    dev = gen.device                      # a device generator draws the full-size (12 B / 20 B parameter) sets in seconds
    w[name + '.weight'] = (torch.randn(out_f, in_f, generator=gen, device=dev) * std).to(dtype)
    w[name + '.bias'] = (torch.randn(out_f, generator=gen, device=dev) * bias_std).to(dtype)


def make_flux_weights(cfg: FluxCfg, seed: int = 0, dtype=torch.bfloat16, teacher_head: bool = False,
                      device='cpu') -> Dict[str, Tensor]:
    """Random FLUX-architecture weights with the diffusers key names (N(0,0.02^2); RMSNorm weights
    1+N(0,0.02^2); modulation / gate linears N(0,0.02^2) with a larger bias so gates are not ~0)."""
    g = torch.Generator(device=device).manual_seed(seed)
    D, w = cfg.dim, {}
    _lin_init(w, 'x_embedder', D, cfg.in_channels, g, std=0.1, dtype=dtype)
    _lin_init(w, 'context_embedder', D, cfg.joint_dim, g, dtype=dtype)
    names = ['timestep_embedder', 'text_embedder'] + (['guidance_embedder'] if cfg.guidance_embeds else [])
    for nm in names:
        in1 = cfg.pooled_dim if nm == 'text_embedder' else 256
        _lin_init(w, f'time_text_embed.{nm}.linear_1', D, in1, g, std=0.05, dtype=dtype)
        _lin_init(w, f'time_text_embed.{nm}.linear_2', D, D, g, std=0.03, dtype=dtype)

    def rmsw(name):
        w[name] = (1 + 0.02 * torch.randn(cfg.head_dim, generator=g, device=g.device)).to(dtype)
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


def make_qwen_weights(cfg: QwenCfg, seed: int = 0, dtype=torch.bfloat16, device='cpu') -> Dict[str, Tensor]:
    g = torch.Generator(device=device).manual_seed(seed)
    D, w = cfg.dim, {}
    _lin_init(w, 'img_in', D, cfg.in_channels, g, std=0.1, dtype=dtype)
    _lin_init(w, 'txt_in', D, cfg.joint_dim, g, dtype=dtype)
    w['txt_norm.weight'] = (1 + 0.02 * torch.randn(cfg.joint_dim, generator=g, device=g.device)).to(dtype)
    _lin_init(w, 'time_text_embed.timestep_embedder.linear_1', D, 256, g, std=0.05, dtype=dtype)
    _lin_init(w, 'time_text_embed.timestep_embedder.linear_2', D, D, g, std=0.03, dtype=dtype)
    for i in range(cfg.num_layers):
        p = f'transformer_blocks.{i}.'
        _lin_init(w, p + 'img_mod.1', 6 * D, D, g, bias_std=0.3, dtype=dtype)
        _lin_init(w, p + 'txt_mod.1', 6 * D, D, g, bias_std=0.3, dtype=dtype)
        for nm in ('to_q', 'to_k', 'to_v', 'add_q_proj', 'add_k_proj', 'add_v_proj', 'to_out.0', 'to_add_out'):
            _lin_init(w, p + 'attn.' + nm, D, D, g, dtype=dtype)
        for nm in ('norm_q', 'norm_k', 'norm_added_q', 'norm_added_k'):
            w[p + f'attn.{nm}.weight'] = (1 + 0.02 * torch.randn(cfg.head_dim, generator=g, device=g.device)).to(dtype)
        for ff in ('img_mlp', 'txt_mlp'):
            _lin_init(w, p + ff + '.net.0.proj', cfg.mlp_ratio * D, D, g, dtype=dtype)
            _lin_init(w, p + ff + '.net.2', D, cfg.mlp_ratio * D, g, dtype=dtype)
    _lin_init(w, 'norm_out.linear', 2 * D, D, g, bias_std=0.1, dtype=dtype)
    K, C, L = cfg.num_gaussians, cfg.in_channels, cfg.logweights_channels
    _lin_init(w, 'proj_out_means', K * C, D, g, dtype=dtype)
    _lin_init(w, 'proj_out_logweights', K * L, D, g, dtype=dtype)
    _lin_init(w, 'proj_out_loggamma', (K - 1) * L, D, g, bias_std=0.5, dtype=dtype)
    return w
