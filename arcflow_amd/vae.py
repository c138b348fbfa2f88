"""AutoencoderKL decoder on the MI355X engine: the ``vae.decode`` step that follows the ArcFlow loop
(reference lakonlab/pipelines/arcflux_pipeline.py:531-534, wrapper lakonlab/models/architecture/diffusers/pretrained.py:22-149).

MI355X-first layout: NHWC bf16 activations on zero-bordered grids; every 3x3 convolution is an implicit GEMM on the
8-phase MFMA kernel (tap = row shift, no im2col, residual add fused in the epilogue); GroupNorm(32)+SiLU, nearest
upsample, softmax and the layout conversions are single-pass streaming kernels; the mid-block attention (one head,
dim 512, 16 384 tokens at 1024^2) is two MFMA GEMMs around an fp32 row softmax.  Weights are re-laid once at load:
conv [Cout,Cin,3,3] -> [Cout][tap][Cin] (K-contiguous), to_q|to_k|to_v stacked.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, Sequence

import torch

from . import _lib, ops


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Grid:
    """Zero-bordered NHWC activation [(H+2)*(W+2), C] with guard rows on both sides (the shifted taps of the implicit
    GEMM read up to W+3 rows outside the grid; what they produce lands on border rows, which the epilogue zeroes)."""

    def __init__(self, H: int, W: int, Cn: int, device, buf: torch.Tensor = None):
        self.H, self.W, self.C = H, W, Cn
        self.guard = W + 3
        rows = (H + 2) * (W + 2)
        self.buf = torch.zeros(rows + 2 * self.guard, Cn, dtype=torch.bfloat16, device=device) if buf is None else buf
        self.t = self.buf[self.guard:self.guard + rows]
        self.stats = None           # GroupNorm partial sums of this grid, left by the convolution that produced it (afx_conv3x3_bf16_stats)


class _GridPool:
    """Recycles activation grids between layers and decodes.  Every producer kernel rewrites the whole grid INCLUDING its
    zero border (conv / linear epilogues, norm kernels, upsample), and nothing ever writes the guard rows, so a recycled
    buffer needs no re-zeroing.  A ring of 6 buffers per shape covers the longest live range of a ResNet block (input kept
    for the residual while norm1 / conv1 / norm2 / conv2 outputs are produced)."""
    RING = 6

    def __init__(self, device):
        self.dev = device
        self.rings = {}

    def grid(self, H: int, W: int, Cn: int) -> _Grid:
        ring = self.rings.setdefault((H, W, Cn), [[], 0])
        if len(ring[0]) < self.RING:
            g = _Grid(H, W, Cn, self.dev)
            ring[0].append(g.buf)
            return g
        buf = ring[0][ring[1] % self.RING]
        ring[1] += 1
        return _Grid(H, W, Cn, self.dev, buf)


def phase_weights(wp9: torch.Tensor, cin: int) -> torch.Tensor:
    """[Cout, 9 * cin] (tap-major 3x3 kernel, the layout of afx_conv3x3_bf16) -> [4, Cout, 4 * cin]: the four 2x2 kernels of
    conv3x3(nearest-2x upsample(.)) acting on the LOW-resolution grid.  Output pixel (2y + py, 2x + px) reads source rows {y - 1 + py, y + py}:
    for py = 0 the taps dy = -1 | {0, +1} fall on them, for py = 1 the taps {-1, 0} | +1 (the same in x); taps on one source pixel add up
    (in fp32, rounded once to bf16).  Phase index 2 py + px, tap index 2 ty + tx: the layout afx_upconv3x3_bf16 expects."""
    co = wp9.shape[0]
    w = wp9.float().reshape(co, 3, 3, cin)
    rows = {0: ([0], [1, 2]), 1: ([0, 1], [2])}                   # phase -> 3x3 tap indices landing on source row 0 / 1 of the 2x2 footprint
    out = torch.zeros(4, co, 2, 2, cin, dtype=torch.float32, device=wp9.device)
    for py in (0, 1):
        for px in (0, 1):
            for ty in (0, 1):
                for tx in (0, 1):
                    acc = 0
                    for a in rows[py][ty]:
                        for b in rows[px][tx]:
                            acc = acc + w[:, a, b]
                    out[2 * py + px, :, ty, tx] = acc
    return out.reshape(4, co, 4 * cin).to(torch.bfloat16).contiguous()


def _upconv(lib, pool, x: "_Grid", w4: torch.Tensor, b: torch.Tensor) -> "_Grid":
    """conv3x3(nearest-2x upsample(x)) + b through afx_upconv3x3_bf16: the upsampled grid is never written."""
    y = pool.grid(2 * x.H, 2 * x.W, w4.shape[1])
    _lib.check(lib.afx_upconv3x3_bf16(_p(x.t), _p(w4), _p(b), _p(y.t), x.H, x.W, x.C, w4.shape[1], _s()))
    return y


def _single_head_attention(lib, xn: _Grid, x: _Grid, w_qkv, b_qkv, w_out, b_out, scale: float) -> _Grid:
    """x + proj(softmax(q k^T * scale) v) over the H*W pixels of a grid (xn = normalised x): two MFMA GEMMs around an fp32
    row softmax.  The token count is padded to a multiple of 64 for the GEMM contraction; padded keys get P = 0."""
    H, W, Cn = x.H, x.W, x.C
    N = H * W
    dev = x.t.device
    Np = (N + 63) // 64 * 64
    xc = torch.zeros(Np, Cn, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.afx_interior_nhwc(_p(xn.t), _p(xc), None, H, W, Cn, 0, _s()))
    qkv = ops.linear(xc, w_qkv, b_qkv)                                                       # [Np, 3C]
    q, k, v = qkv[:, :Cn], qkv[:, Cn:2 * Cn], qkv[:, 2 * Cn:]
    s = ops.linear_f32out(q, k)                                                              # [Np, Np] fp32 logits
    pm = torch.zeros(Np, Np, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.afx_softmax_rows_f32(_p(s), s.stride(0), _p(pm), Np, N, N, scale, _s()))
    o = ops.linear(pm, ops.transpose(v))                                                     # [N, C] = P V
    oc = ops.linear(o, w_out, b_out)
    y = _Grid(H, W, Cn, dev)
    _lib.check(lib.afx_interior_nhwc(_p(y.t), _p(oc), _p(x.t), H, W, Cn, 1, _s()))
    return y


class AutoencoderKLDecoder:
    def __init__(self, state_dict: Dict[str, torch.Tensor], block_out_channels: Sequence[int] = (128, 256, 512, 512),
                 norm_num_groups: int = 32, layers_per_block: int = 2, scaling_factor: float = 0.3611,
                 shift_factor: float = 0.1159, device='cuda'):
        self.lib = _lib.load()
        self.dev = torch.device(device)
        self.groups, self.lpb = norm_num_groups, layers_per_block
        self.rev = list(reversed(block_out_channels))
        self.scaling_factor, self.shift_factor = scaling_factor, shift_factor
        self.config = type('cfg', (), dict(scaling_factor=scaling_factor, shift_factor=shift_factor))()
        self.w: Dict[str, torch.Tensor] = {}
        sd = state_dict
        for k in [k for k in sd if k.startswith('decoder.') and k.endswith('.weight')]:
            name = k[:-len('.weight')]
            wt, b = sd[k], sd.get(name + '.bias')
            if wt.dim() == 4 and wt.shape[-1] == 3:                       # 3x3 conv -> [Cout][tap][Cin], K-contiguous
                co, ci = wt.shape[:2]
                cip, cop = max(64, (ci + 63) // 64 * 64), (co + 7) // 8 * 8
                wp = torch.zeros(cop, 9, cip, dtype=torch.bfloat16)
                wp[:co, :, :ci] = wt.permute(0, 2, 3, 1).reshape(co, 9, ci).to(torch.bfloat16)
                bp = torch.zeros(cop, dtype=torch.bfloat16)
                bp[:co] = b.to(torch.bfloat16)
                self.w[name + '.weight'], self.w[name + '.bias'] = wp.reshape(cop, 9 * cip).to(self.dev), bp.to(self.dev)
            elif wt.dim() == 4:                                           # 1x1 conv_shortcut -> linear
                self.w[name + '.weight'] = wt.reshape(wt.shape[0], wt.shape[1]).to(self.dev, torch.bfloat16).contiguous()
                self.w[name + '.bias'] = b.to(self.dev, torch.bfloat16)
            elif wt.dim() == 2:
                self.w[name + '.weight'] = wt.to(self.dev, torch.bfloat16).contiguous()
                self.w[name + '.bias'] = b.to(self.dev, torch.bfloat16)
            else:                                                         # GroupNorm affine
                self.w[name + '.weight'] = wt.to(self.dev, torch.float32)
                self.w[name + '.bias'] = b.to(self.dev, torch.float32)
        a = 'decoder.mid_block.attentions.0.'
        self.w[a + 'qkv.weight'] = torch.cat([self.w[a + n + '.weight'] for n in ('to_q', 'to_k', 'to_v')]).contiguous()
        self.w[a + 'qkv.bias'] = torch.cat([self.w[a + n + '.bias'] for n in ('to_q', 'to_k', 'to_v')]).contiguous()
        # nearest-2x upsample folded into the upsamplers' convolutions (four 2x2 phase kernels on the low-resolution grid)
        self._fold_up = bool(self.lib.afx_conv_stats_available()) and os.environ.get('AFX_VAE_FOLD_UPSAMPLE', '1') != '0'     # (0: A/B runs)
        if self._fold_up:
            for k in [k for k in self.w if '.upsamplers.0.conv.weight' in k]:
                self.w[k[:-len('.weight')] + '.weight4'] = phase_weights(self.w[k], self.w[k].shape[1] // 9)
        self._stats = torch.zeros(int(self.lib.afx_groupnorm_ws_bytes(2048, self.groups)) // 8, dtype=torch.float64, device=self.dev)   # afx_groupnorm_nhwc's scratch, sized by the library
        self._pool = _GridPool(self.dev)
        # GroupNorm sums out of the producing convolution's epilogue: a ring of slotted buffers (a grid's sums live until its norm ran:
        # at most the block input + conv1 output at a time; 4 is generous)
        self._conv_stats = bool(self.lib.afx_conv_stats_available()) and os.environ.get('AFX_VAE_CONV_STATS', '1') != '0'     # (0: A/B runs)
        self._stat_ring = [torch.zeros(64 * 2 * self.groups, dtype=torch.float64, device=self.dev) for _ in range(4)]
        self._stat_i = 0

    # ------------------------------------------------------------------ primitives on grids
    def _conv(self, name: str, x: _Grid, cout: int, res: _Grid = None, stats: bool = True) -> _Grid:
        """stats: the output feeds a GroupNorm -> its sums come out of the GEMM epilogue (no statistics pass over the grid later)."""
        w, b = self.w[name + '.weight'], self.w[name + '.bias']
        co = w.shape[0]
        y = self._pool.grid(x.H, x.W, co)
        gs = co // self.groups if co % self.groups == 0 else 0
        if stats and self._conv_stats and co <= 128 and gs >= 4 and gs % 4 == 0 and (gs <= 8 or gs % 8 == 0):
            y.stats = self._stat_ring[self._stat_i % len(self._stat_ring)]
            self._stat_i += 1
            _lib.check(self.lib.afx_conv3x3_bf16_stats(_p(x.t), _p(w), _p(b), _p(y.t), x.H, x.W, x.C, co, None if res is None else _p(res.t),
                                                       _p(y.stats), self.groups, _s()))
        else:
            _lib.check(self.lib.afx_conv3x3_bf16(_p(x.t), _p(w), _p(b), _p(y.t), x.H, x.W, x.C, co, None if res is None else _p(res.t), _s()))
        return y

    def _gn(self, name: str, x: _Grid, act: bool) -> _Grid:
        y = self._pool.grid(x.H, x.W, x.C)
        g, b = self.w[name + '.weight'], self.w[name + '.bias']
        if x.stats is not None:
            _lib.check(self.lib.afx_groupnorm_nhwc_from_stats(_p(x.t), _p(y.t), _p(x.stats), _p(self._stats), x.H, x.W, x.C, self.groups,
                                                              _p(g), _p(b), 1e-6, int(act), _s()))
        else:
            _lib.check(self.lib.afx_groupnorm_nhwc(_p(x.t), _p(y.t), _p(self._stats), x.H, x.W, x.C, self.groups, _p(g), _p(b), 1e-6, int(act), _s()))
        return y

    def _resnet(self, p: str, x: _Grid) -> _Grid:
        h = self._conv(p + 'conv1', self._gn(p + 'norm1', x, True), 0)
        skip = x
        if p + 'conv_shortcut.weight' in self.w:
            skip = self._pool.grid(x.H, x.W, self.w[p + 'conv_shortcut.weight'].shape[0])
            ops.linear(x.t, self.w[p + 'conv_shortcut.weight'], self.w[p + 'conv_shortcut.bias'], out=skip.t)
        return self._conv(p + 'conv2', self._gn(p + 'norm2', h, True), 0, res=skip)

    def _attention(self, p: str, x: _Grid) -> _Grid:
        xn = self._gn(p + 'group_norm', x, False)
        return _single_head_attention(self.lib, xn, x, self.w[p + 'qkv.weight'], self.w[p + 'qkv.bias'],
                                      self.w[p + 'to_out.0.weight'], self.w[p + 'to_out.0.bias'], x.C ** -0.5)

    # ------------------------------------------------------------------ decode
    @torch.no_grad()
    def decode_tokens(self, tokens: torch.Tensor, hp: int, wp: int) -> torch.Tensor:
        """tokens: packed latents [hp*wp, 64] fp32 (the loop's output for ONE image) -> image [3, 16hp, 16wp] fp32."""
        x = self._pool.grid(2 * hp, 2 * wp, 64)
        _lib.check(self.lib.afx_latent_to_nhwc(_p(tokens.to(self.dev, torch.float32).contiguous()), _p(x.t), hp, wp, 64,
                                               self.scaling_factor, self.shift_factor, _s()))
        x = self._conv('decoder.conv_in', x, 0)
        x = self._resnet('decoder.mid_block.resnets.0.', x)
        x = self._attention('decoder.mid_block.attentions.0.', x)
        x = self._resnet('decoder.mid_block.resnets.1.', x)
        n = len(self.rev)
        for i in range(n):
            for j in range(self.lpb + 1):
                x = self._resnet(f'decoder.up_blocks.{i}.resnets.{j}.', x)
            if i < n - 1:
                un = f'decoder.up_blocks.{i}.upsamplers.0.conv'
                if self._fold_up:
                    x = _upconv(self.lib, self._pool, x, self.w[un + '.weight4'], self.w[un + '.bias'])
                else:
                    up = self._pool.grid(2 * x.H, 2 * x.W, x.C)
                    _lib.check(self.lib.afx_upsample2x_nhwc(_p(x.t), _p(up.t), x.H, x.W, x.C, _s()))
                    x = self._conv(un, up, 0)
        x = self._conv('decoder.conv_out', self._gn('decoder.conv_norm_out', x, True), 0, stats=False)
        img = torch.empty(3, x.H, x.W, dtype=torch.float32, device=self.dev)
        _lib.check(self.lib.afx_nhwc_to_image(_p(x.t), _p(img), x.H, x.W, x.C, _s()))
        return img

    def decode_packed(self, latents: torch.Tensor, hp: int, wp: int) -> torch.Tensor:
        """[B, hp*wp, 64] packed latents -> [B, 3, 16hp, 16wp]."""
        return torch.stack([self.decode_tokens(latents[b], hp, wp) for b in range(latents.shape[0])])


def _c64(c: int) -> int:
    return (c + 63) // 64 * 64


class AutoencoderKLQwenImageDecoder:
    """Decoder of AutoencoderKLQwenImage for single images (reference arcqwen_pipeline.py:470-481 and
    lakonlab/models/architecture/diffusers/pretrained.py:142-149).  With one latent frame the causal 3-D convolutions
    see two zero frames in front, so each reduces to its LAST temporal tap as a 2-D kernel and the temporal ``time_conv``
    of the 3-D upsamplers is never reached; those 2-D kernels run as implicit GEMMs like the FLUX decoder's.  Channel
    counts that are not multiples of 64 (96) live on grids padded to the next multiple (zero weights / gamma there).
    The per-channel latent un-normalisation and the 1x1x1 ``post_quant_conv`` are folded into the unpack kernel."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], latents_mean: Sequence[float], latents_std: Sequence[float],
                 dim_mult: Sequence[int] = (1, 2, 4, 4), num_res_blocks: int = 2, z_dim: int = 16, device='cuda'):
        assert z_dim == 16, 'the packed-latent layout of the pipelines has 16 latent channels'
        self.lib = _lib.load()
        self.dev = torch.device(device)
        self.n_up, self.nrb = len(dim_mult), num_res_blocks
        self.latents_mean, self.latents_std = list(latents_mean), list(latents_std)
        self.config = type('cfg', (), dict(latents_mean=self.latents_mean, latents_std=self.latents_std, z_dim=z_dim))()
        self.w: Dict[str, torch.Tensor] = {}
        self.creal: Dict[str, int] = {}
        sd = {k: v.detach().cpu() for k, v in state_dict.items()}          # repacked on the host (a few hundred MB, once), then uploaded
        for k in [k for k in sd if k.startswith('decoder.') and k.endswith('.weight') and 'time_conv' not in k]:
            name = k[:-len('.weight')]
            wt, b = sd[k].float(), sd[name + '.bias'].float()
            if wt.dim() == 5:
                wt = wt[:, :, -1]                                         # causal: only the last temporal tap sees the frame
            co, ci = wt.shape[:2]
            cop = 8 if name == 'decoder.conv_out' else _c64(co)
            cip = _c64(ci)
            bp = torch.zeros(cop, dtype=torch.bfloat16)
            bp[:co] = b.to(torch.bfloat16)
            if wt.shape[-1] == 3:                                         # 3x3 -> [Cout][tap][Cin], K-contiguous
                wp = torch.zeros(cop, 9, cip, dtype=torch.bfloat16)
                wp[:co, :, :ci] = wt.permute(0, 2, 3, 1).reshape(co, 9, ci).to(torch.bfloat16)
                self.w[name + '.weight'] = wp.reshape(cop, 9 * cip).to(self.dev)
            else:                                                         # 1x1 -> linear
                wp = torch.zeros(cop, cip, dtype=torch.bfloat16)
                wp[:co, :ci] = wt.reshape(co, ci).to(torch.bfloat16)
                self.w[name + '.weight'] = wp.to(self.dev)
            self.w[name + '.bias'] = bp.to(self.dev)
            self.creal[name] = co
        for k in [k for k in sd if k.startswith('decoder.') and k.endswith('.gamma')]:
            gm = sd[k].float().flatten()
            gp = torch.zeros(_c64(gm.numel()), dtype=torch.float32)
            gp[:gm.numel()] = gm
            self.w[k], self.creal[k] = gp.to(self.dev), gm.numel()
        # v = post_quant_conv(lat * std + mean) = (Wq diag(std)) lat + (Wq mean + bq)
        wq = sd['post_quant_conv.weight'].float().reshape(16, 16)
        std, mean = torch.tensor(self.latents_std, dtype=torch.float32), torch.tensor(self.latents_mean, dtype=torch.float32)
        self._pool = _GridPool(self.dev)
        self._fold_up = bool(self.lib.afx_conv_stats_available()) and os.environ.get('AFX_VAE_FOLD_UPSAMPLE', '1') != '0'
        if self._fold_up:
            for k in [k for k in self.w if '.upsamplers.0.resample.1.weight' in k]:
                self.w[k[:-len('.weight')] + '.weight4'] = phase_weights(self.w[k], self.w[k].shape[1] // 9)
        self._A = (wq * std[None, :]).contiguous().to(self.dev)
        self._b = (wq @ mean + sd['post_quant_conv.bias'].float()).contiguous().to(self.dev)

    def _conv(self, name: str, x: _Grid, res: _Grid = None) -> _Grid:
        w, b = self.w[name + '.weight'], self.w[name + '.bias']
        y = self._pool.grid(x.H, x.W, w.shape[0])
        _lib.check(self.lib.afx_conv3x3_bf16(_p(x.t), _p(w), _p(b), _p(y.t), x.H, x.W, x.C, w.shape[0],
                                             None if res is None else _p(res.t), _s()))
        return y

    def _norm(self, name: str, x: _Grid, act: bool) -> _Grid:
        y = self._pool.grid(x.H, x.W, x.C)
        _lib.check(self.lib.afx_rmsnorm_nhwc(_p(x.t), _p(y.t), x.t.shape[0], x.C, self.creal[name + '.gamma'],
                                             _p(self.w[name + '.gamma']), int(act), _s()))
        return y

    def _resnet(self, p: str, x: _Grid) -> _Grid:
        skip = x
        if p + 'conv_shortcut.weight' in self.w:
            skip = self._pool.grid(x.H, x.W, self.w[p + 'conv_shortcut.weight'].shape[0])
            ops.linear(x.t, self.w[p + 'conv_shortcut.weight'], self.w[p + 'conv_shortcut.bias'], out=skip.t)
        h = self._conv(p + 'conv1', self._norm(p + 'norm1', x, True))
        return self._conv(p + 'conv2', self._norm(p + 'norm2', h, True), res=skip)

    @torch.no_grad()
    def decode_tokens(self, tokens: torch.Tensor, hp: int, wp: int) -> torch.Tensor:
        """tokens: packed latents [hp*wp, 64] fp32 of ONE image -> image [3, 16hp, 16wp] fp32 in [-1, 1]."""
        x = self._pool.grid(2 * hp, 2 * wp, 64)
        _lib.check(self.lib.afx_latent_to_nhwc_affine(_p(tokens.to(self.dev, torch.float32).contiguous()), _p(x.t), hp, wp, 64,
                                                      _p(self._A), _p(self._b), _s()))
        x = self._conv('decoder.conv_in', x)
        x = self._resnet('decoder.mid_block.resnets.0.', x)
        a = 'decoder.mid_block.attentions.0.'
        x = _single_head_attention(self.lib, self._norm(a + 'norm', x, False), x, self.w[a + 'to_qkv.weight'], self.w[a + 'to_qkv.bias'],
                                   self.w[a + 'proj.weight'], self.w[a + 'proj.bias'], self.creal[a + 'norm.gamma'] ** -0.5)
        x = self._resnet('decoder.mid_block.resnets.1.', x)
        for i in range(self.n_up):
            for j in range(self.nrb + 1):
                x = self._resnet(f'decoder.up_blocks.{i}.resnets.{j}.', x)
            if i != self.n_up - 1:
                un = f'decoder.up_blocks.{i}.upsamplers.0.resample.1'
                if self._fold_up:
                    x = _upconv(self.lib, self._pool, x, self.w[un + '.weight4'], self.w[un + '.bias'])
                else:
                    up = self._pool.grid(2 * x.H, 2 * x.W, x.C)
                    _lib.check(self.lib.afx_upsample2x_nhwc(_p(x.t), _p(up.t), x.H, x.W, x.C, _s()))
                    x = self._conv(un, up)
        x = self._conv('decoder.conv_out', self._norm('decoder.norm_out', x, True))
        img = torch.empty(3, x.H, x.W, dtype=torch.float32, device=self.dev)
        _lib.check(self.lib.afx_nhwc_to_image(_p(x.t), _p(img), x.H, x.W, x.C, _s()))
        return img.clamp_(-1.0, 1.0)

    def decode_packed(self, latents: torch.Tensor, hp: int, wp: int) -> torch.Tensor:
        return torch.stack([self.decode_tokens(latents[b], hp, wp) for b in range(latents.shape[0])])
