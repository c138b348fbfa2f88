"""CPU oracle (torch fp32) of the AutoencoderKLQwenImage *decoder* for one frame (T = 1), the step after the ArcFlow
loop in the Qwen pipeline (reference call site lakonlab/pipelines/arcqwen_pipeline.py:470-481, training wrapper
lakonlab/models/architecture/diffusers/pretrained.py:105-149; the class is imported from diffusers==0.35.1 -- an
absent third-party dependency, so this restates its published decoder):

    post_quant_conv (causal 1x1x1) -> conv_in (causal 3x3x3) -> mid block (ResBlock, single-head attention, ResBlock)
    -> 4 up blocks of 3 ResBlocks (dims 384, 384, 192, 96; nearest-exact 2x upsample + Conv2d(dim, dim/2, 3) after the
    first three; the temporal ``time_conv`` of the 3-D upsamplers is skipped for the first frame) -> RMS norm -> SiLU ->
    conv_out (causal 3x3x3) -> clamp(-1, 1)

with causal convolutions padding two zero frames in FRONT (so at T = 1 only the last temporal tap sees data) and
``RMS_norm(x) = F.normalize(x, dim=channels) * sqrt(C) * gamma``.  The convolutions are evaluated as real conv3d on the
5-D weights, so the engine's "last temporal tap as a 2-D kernel" reduction is checked, not assumed.

TEST INFRASTRUCTURE ONLY.  Parity status: **parity unpinned** (no reference vector exists for the VAE).
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def causal_conv3d(w, name, x):
    wt = w[name + '.weight'].float()
    kt, kh, kw = wt.shape[2:]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, kt - 1, 0))
    return F.conv3d(x, wt, w[name + '.bias'].float())


def rms_norm(w, name, x):
    g = w[name + '.gamma'].float().reshape(1, -1, *([1] * (x.dim() - 2)))
    return F.normalize(x, dim=1) * (x.shape[1] ** 0.5) * g


def res_block(w, p, x):
    h = causal_conv3d(w, p + 'conv_shortcut', x) if p + 'conv_shortcut.weight' in w else x
    x = causal_conv3d(w, p + 'conv1', F.silu(rms_norm(w, p + 'norm1', x)))
    x = causal_conv3d(w, p + 'conv2', F.silu(rms_norm(w, p + 'norm2', x)))
    return x + h


def attention_block(w, p, x):
    b, c, t, hh, ww = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww)
    y = rms_norm(w, p + 'norm', y)
    qkv = F.conv2d(y, w[p + 'to_qkv.weight'].float(), w[p + 'to_qkv.bias'].float())
    qkv = qkv.reshape(b * t, 1, c * 3, hh * ww).permute(0, 1, 3, 2)
    q, k, v = qkv.chunk(3, dim=-1)
    a = torch.softmax(q @ k.transpose(-1, -2) / c ** 0.5, dim=-1) @ v
    a = a.squeeze(1).permute(0, 2, 1).reshape(b * t, c, hh, ww)
    a = F.conv2d(a, w[p + 'proj.weight'].float(), w[p + 'proj.bias'].float())
    return x + a.reshape(b, t, c, hh, ww).permute(0, 2, 1, 3, 4)


def upsample(w, p, x):
    b, c, t, hh, ww = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww)
    y = F.interpolate(y, scale_factor=(2.0, 2.0), mode='nearest-exact')
    y = F.conv2d(y, w[p + 'resample.1.weight'].float(), w[p + 'resample.1.bias'].float(), padding=1)
    return y.reshape(b, t, y.shape[1], 2 * hh, 2 * ww).permute(0, 2, 1, 3, 4)


def decode(w: Dict[str, Tensor], z: Tensor, dim_mult: Sequence[int] = (1, 2, 4, 4), num_res_blocks: int = 2) -> Tensor:
    """z [B, 16, H, W] (already un-normalised: lat * latents_std + latents_mean) -> image [B, 3, 8H, 8W] in [-1, 1]."""
    x = z.float()[:, :, None]
    x = causal_conv3d(w, 'post_quant_conv', x)
    x = causal_conv3d(w, 'decoder.conv_in', x)
    x = res_block(w, 'decoder.mid_block.resnets.0.', x)
    x = attention_block(w, 'decoder.mid_block.attentions.0.', x)
    x = res_block(w, 'decoder.mid_block.resnets.1.', x)
    n = len(dim_mult)
    for i in range(n):
        for j in range(num_res_blocks + 1):
            x = res_block(w, f'decoder.up_blocks.{i}.resnets.{j}.', x)
        if i != n - 1:
            x = upsample(w, f'decoder.up_blocks.{i}.upsamplers.0.', x)
    x = causal_conv3d(w, 'decoder.conv_out', F.silu(rms_norm(w, 'decoder.norm_out', x)))
    return x[:, :, 0].clamp(-1.0, 1.0)


def make_decoder_weights(dim: int = 96, z_dim: int = 16, dim_mult: Sequence[int] = (1, 2, 4, 4), num_res_blocks: int = 2,
                         temporal_upsample: Sequence[bool] = (False, True, True), seed: int = 0, std: float = 0.03) -> Dict[str, Tensor]:
    """Random weights with the key names / shapes of the diffusers class (decoder + post_quant_conv)."""
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, Tensor] = {}

    def conv3(name, co, ci, k):
        w[name + '.weight'] = torch.randn(co, ci, k, k, k, generator=g) * (std if k == 3 else std * 3)
        w[name + '.bias'] = torch.randn(co, generator=g) * 0.02

    def conv2(name, co, ci, k):
        w[name + '.weight'] = torch.randn(co, ci, k, k, generator=g) * (std if k == 3 else std * 2)
        w[name + '.bias'] = torch.randn(co, generator=g) * 0.02

    def norm(name, c, images):
        w[name + '.gamma'] = (1.0 + 0.1 * torch.randn(c, generator=g)).reshape((c, 1, 1) if images else (c, 1, 1, 1))

    def res(p, ci, co):
        norm(p + 'norm1', ci, False); conv3(p + 'conv1', co, ci, 3)
        norm(p + 'norm2', co, False); conv3(p + 'conv2', co, co, 3)
        if ci != co:
            conv3(p + 'conv_shortcut', co, ci, 1)

    dims = [dim * u for u in [dim_mult[-1]] + list(dim_mult[::-1])]
    conv3('post_quant_conv', z_dim, z_dim, 1)
    conv3('decoder.conv_in', dims[0], z_dim, 3)
    res('decoder.mid_block.resnets.0.', dims[0], dims[0])
    a = 'decoder.mid_block.attentions.0.'
    norm(a + 'norm', dims[0], True); conv2(a + 'to_qkv', dims[0] * 3, dims[0], 1); conv2(a + 'proj', dims[0], dims[0], 1)
    res('decoder.mid_block.resnets.1.', dims[0], dims[0])
    for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
        if i > 0:
            ci = ci // 2
        cur = ci
        for j in range(num_res_blocks + 1):
            res(f'decoder.up_blocks.{i}.resnets.{j}.', cur, co)
            cur = co
        if i != len(dim_mult) - 1:
            conv2(f'decoder.up_blocks.{i}.upsamplers.0.resample.1', co // 2, co, 3)
            if temporal_upsample[i]:
                conv3(f'decoder.up_blocks.{i}.upsamplers.0.time_conv', co * 2, co, 1)
                w[f'decoder.up_blocks.{i}.upsamplers.0.time_conv.weight'] = torch.randn(co * 2, co, 3, 1, 1, generator=g) * std
    norm('decoder.norm_out', dims[-1], False)
    conv3('decoder.conv_out', 3, dims[-1], 3)
    return w
