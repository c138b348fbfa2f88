"""CPU oracle (torch fp32) of the AutoencoderKL *decoder* the FLUX pipeline calls after the denoising loop
(reference call site lakonlab/pipelines/arcflux_pipeline.py:531-534; class imported from diffusers==0.35.1,
arcflux_pipeline.py:18 -- absent third-party dependency, so this restates diffusers' published Decoder:
conv_in -> UNetMidBlock2D(ResnetBlock2D, single-head Attention, ResnetBlock2D) -> 4 x UpDecoderBlock2D
(3 ResnetBlock2D each, nearest-2x Upsample2D + conv on the first three) -> GroupNorm -> SiLU -> conv_out).

TEST INFRASTRUCTURE ONLY.  Parity status: **parity unpinned** (no reference vector exists for the VAE).
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _gn(w, name, x, groups, eps=1e-6):
    return F.group_norm(x, groups, w[name + '.weight'].float(), w[name + '.bias'].float(), eps)


def _conv(w, name, x, pad=1):
    return F.conv2d(x, w[name + '.weight'].float(), w[name + '.bias'].float(), padding=pad)


def resnet(w, p, x, groups):
    h = _conv(w, p + 'conv1', F.silu(_gn(w, p + 'norm1', x, groups)))
    h = _conv(w, p + 'conv2', F.silu(_gn(w, p + 'norm2', h, groups)))
    if p + 'conv_shortcut.weight' in w:
        x = _conv(w, p + 'conv_shortcut', x, pad=0)
    return x + h


def mid_attention(w, p, x, groups):
    b, c, hh, ww = x.shape
    t = _gn(w, p + 'group_norm', x, groups).reshape(b, c, hh * ww).transpose(1, 2)
    q = F.linear(t, w[p + 'to_q.weight'].float(), w[p + 'to_q.bias'].float())
    k = F.linear(t, w[p + 'to_k.weight'].float(), w[p + 'to_k.bias'].float())
    v = F.linear(t, w[p + 'to_v.weight'].float(), w[p + 'to_v.bias'].float())
    a = torch.softmax(q @ k.transpose(1, 2) / c ** 0.5, dim=-1) @ v
    o = F.linear(a, w[p + 'to_out.0.weight'].float(), w[p + 'to_out.0.bias'].float())
    return x + o.transpose(1, 2).reshape(b, c, hh, ww)


def decode(w: Dict[str, Tensor], z: Tensor, block_out_channels: Sequence[int] = (128, 256, 512, 512), groups: int = 32,
           layers_per_block: int = 2) -> Tensor:
    """z [B, 16, H, W] (already un-scaled: lat / scaling_factor + shift_factor) -> image [B, 3, 8H, 8W]."""
    x = _conv(w, 'decoder.conv_in', z.float())
    x = resnet(w, 'decoder.mid_block.resnets.0.', x, groups)
    x = mid_attention(w, 'decoder.mid_block.attentions.0.', x, groups)
    x = resnet(w, 'decoder.mid_block.resnets.1.', x, groups)
    n = len(block_out_channels)
    for i in range(n):
        for j in range(layers_per_block + 1):
            x = resnet(w, f'decoder.up_blocks.{i}.resnets.{j}.', x, groups)
        if i < n - 1:
            x = F.interpolate(x, scale_factor=2.0, mode='nearest')
            x = _conv(w, f'decoder.up_blocks.{i}.upsamplers.0.conv', x)
    x = F.silu(_gn(w, 'decoder.conv_norm_out', x, groups))
    return _conv(w, 'decoder.conv_out', x)


def make_decoder_weights(block_out_channels=(128, 256, 512, 512), latent_channels=16, layers_per_block=2, seed=0,
                         dtype=torch.bfloat16) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, Tensor] = {}

    def conv(name, co, ci, k=3):
        w[name + '.weight'] = (torch.randn(co, ci, k, k, generator=g) * (1.2 / (ci * k * k) ** 0.5)).to(dtype)
        w[name + '.bias'] = (torch.randn(co, generator=g) * 0.05).to(dtype)

    def norm(name, c):
        w[name + '.weight'] = (1 + 0.1 * torch.randn(c, generator=g)).to(dtype)
        w[name + '.bias'] = (0.1 * torch.randn(c, generator=g)).to(dtype)

    def res(p, ci, co):
        norm(p + 'norm1', ci); conv(p + 'conv1', co, ci); norm(p + 'norm2', co); conv(p + 'conv2', co, co)
        if ci != co:
            conv(p + 'conv_shortcut', co, ci, 1)
    rev = list(reversed(block_out_channels))
    c0 = rev[0]
    conv('decoder.conv_in', c0, latent_channels)
    res('decoder.mid_block.resnets.0.', c0, c0)
    p = 'decoder.mid_block.attentions.0.'
    norm(p + 'group_norm', c0)
    for nm in ('to_q', 'to_k', 'to_v', 'to_out.0'):
        w[p + nm + '.weight'] = (torch.randn(c0, c0, generator=g) * (1.0 / c0 ** 0.5)).to(dtype)
        w[p + nm + '.bias'] = (torch.randn(c0, generator=g) * 0.05).to(dtype)
    res('decoder.mid_block.resnets.1.', c0, c0)
    prev = c0
    for i, co in enumerate(rev):
        for j in range(layers_per_block + 1):
            res(f'decoder.up_blocks.{i}.resnets.{j}.', prev if j == 0 else co, co)
        prev = co
        if i < len(rev) - 1:
            conv(f'decoder.up_blocks.{i}.upsamplers.0.conv', co, co)
    norm('decoder.conv_norm_out', prev)
    conv('decoder.conv_out', 3, prev)
    return w
