"""Python handle on one libarcflow_hip context: the MI355X denoiser that stands in for the reference's
``transformer`` module (``_ArcFluxTransformer2DModel`` / ``_ArcQwenImageTransformer2DModel``,
lakonlab/models/architecture/arcflow/arcflux.py:25-257, arcqwen.py:23-174).

torch is used for device memory and the current HIP stream only; all math runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib, rope
from .weights import pack_flux, pack_qwen


@dataclass
class ArcFlowModelOutput:
    """Same fields as the reference's ArcFlowModelOutput (arc_output.py:9-25)."""
    means: torch.Tensor        # [B, N, K, C]
    logweights: torch.Tensor   # [B, N, K, L]   log_softmax over K
    loggammas: torch.Tensor    # [B, N, K-1, L]

    def __getitem__(self, k):
        return getattr(self, k)

    def keys(self):
        return ('means', 'logweights', 'loggammas')

    def items(self):
        return [(k, getattr(self, k)) for k in self.keys()]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class MMDiTEngine:
    """FLUX / Qwen-Image MMDiT trunk + ArcFlow heads on one GPU."""

    def __init__(self, family: str, num_double: int, num_single: int = 0, heads: int = 24, head_dim: int = 128,
                 in_channels: int = 64, joint_dim: int = 4096, pooled_dim: int = 768, guidance_embeds: bool = True,
                 num_gaussians: int = 16, logweights_channels: int = 4, teacher_head: bool = False,
                 axes_dims=(16, 56, 56), device='cuda'):
        assert family in ('flux', 'qwen')
        self.lib = _lib.load()
        self.family = family
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        self.num_double, self.num_single = num_double, num_single
        self.heads, self.head_dim, self.dim = heads, head_dim, heads * head_dim
        self.in_channels, self.joint_dim = in_channels, joint_dim
        self.pooled_dim = pooled_dim if family == 'flux' else 0
        self.guidance_embeds = bool(guidance_embeds) if family == 'flux' else False
        self.num_gaussians, self.logweights_channels = num_gaussians, logweights_channels
        self.teacher_head = teacher_head
        self.axes_dims = tuple(axes_dims)
        desc = _lib.ModelDesc(0 if family == 'flux' else 1, num_double, num_single, heads, head_dim, in_channels,
                              joint_dim, self.pooled_dim, int(self.guidance_embeds), num_gaussians,
                              logweights_channels, int(teacher_head))
        self._ctx = C.c_void_p()
        _lib.check(self.lib.afx_create(C.byref(desc), C.byref(self._ctx)))
        self._weights: Dict[str, torch.Tensor] = {}
        self._ws: Optional[torch.Tensor] = None
        self._ws_key = (0, 0, 0)
        self._rope_cache: Dict[Tuple[int, int, int], Tuple[torch.Tensor, torch.Tensor]] = {}
        self.config = dict(in_channels=in_channels, guidance_embeds=self.guidance_embeds, num_gaussians=num_gaussians)

    def __del__(self):
        try:
            if getattr(self, '_ctx', None) and self._ctx.value:
                self.lib.afx_destroy(self._ctx)
                self._ctx = C.c_void_p()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Fuse + upload a diffusers-keyed state dict (optionally already LoRA-merged) and bind it."""
        if self.family == 'flux':
            packed = pack_flux(sd, self.num_double, self.num_single, self.device, self.num_gaussians,
                               self.in_channels, self.logweights_channels, self.guidance_embeds, self.teacher_head)
        else:
            packed = pack_qwen(sd, self.num_double, self.device, self.num_gaussians, self.in_channels,
                               self.logweights_channels, self.teacher_head)
        self.bind_packed(packed)

    def bind_packed(self, packed: Dict[str, torch.Tensor]) -> None:
        for name, t in packed.items():
            assert t.is_contiguous() and t.device.type == 'cuda', name
            dt = {torch.bfloat16: _lib.AFX_DT_BF16, torch.float32: _lib.AFX_DT_F32, torch.uint8: _lib.AFX_DT_FP8}[t.dtype]
            shape = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(self.lib.afx_bind_weight(self._ctx, name.encode(), _ptr(t), dt, t.dim(), shape))
        self._weights.update(packed)         # keep the storage alive
        _lib.check(self.lib.afx_finalize(self._ctx))

    # ------------------------------------------------------------------ helpers
    def _workspace(self, B: int, N: int, T: int) -> None:
        if self._ws is not None and all(a <= b for a, b in zip((B, N, T), self._ws_key)):
            return
        key = tuple(max(a, b) for a, b in zip((B, N, T), self._ws_key))
        need = self.lib.afx_workspace_bytes(self._ctx, *key)
        if need < 0:
            _lib.check(int(need))
        self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        base = (self._ws.data_ptr() + 255) // 256 * 256
        _lib.check(self.lib.afx_set_workspace(self._ctx, C.c_void_p(base), need))
        self._ws_key = key

    def rope_tables(self, hp: int, wp: int, T: int):
        k = (hp, wp, T)
        if k not in self._rope_cache:
            if self.family == 'flux':
                cs = rope.flux_tables(hp, wp, T, self.axes_dims)
            else:
                cs = rope.qwen_tables(hp, wp, T, self.axes_dims)
            self._rope_cache[k] = tuple(t.to(self.device) for t in cs)
        return self._rope_cache[k]

    def export(self, what: str, dst: torch.Tensor, B: int, N: int, T: int) -> torch.Tensor:
        """Copy an activation of the last forward out of the workspace ('head_in', 'x_final', 'silu_temb',
        'mod_final'); the distillation step needs them for the head / norm_out gradients."""
        _lib.check(self.lib.afx_mmdit_export(self._ctx, what.encode(), _ptr(dst), B, N, T, _stream()))
        return dst

    def set_checkpoint_buffer(self, buf: Optional[torch.Tensor]) -> None:
        """[num_blocks, B*S, D] bf16 tensor that receives every block's input on the next forwards (None = off)."""
        self._ckpt = buf
        _lib.check(self.lib.afx_set_checkpoint_buffer(self._ctx, _ptr(buf)))

    @property
    def n_mod(self) -> int:
        return (self.num_double * 12 + self.num_single * 3 + 2) * self.dim

    # ------------------------------------------------------------------ instrumentation
    def profile(self, on) -> None:
        """False / True: off / an event pair on every GEMM and attention launch; an int N > 1: on one launch in N (sampled)."""
        _lib.check(self.lib.afx_profile_enable(self._ctx, int(on)))

    def profile_read(self, klass: int):
        """(total_ms, launches, algorithmic_flops) of GEMM (klass 0) / attention (klass 1) launches."""
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        _lib.check(self.lib.afx_profile_read(self._ctx, klass, C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value

    # ------------------------------------------------------------------ conditioning of several steps at once
    @torch.no_grad()
    def prepare_steps(self, timesteps, pooled_projections: Optional[torch.Tensor] = None, guidance: Optional[torch.Tensor] = None,
                      batch: int = 1, n_img: int = 1, n_txt: int = 1) -> bool:
        """Compute the AdaLN modulation vectors of ALL the sampler's steps in one pass over the stacked modulation matrix
        (``afx_mmdit_prepare_steps``): ``timesteps`` [nsteps] (shared by the batch) or [nsteps, B] sigmas in [0, 1].  A following
        ``forward(..., prepared_step=k)`` of the same batch then skips its own pass (1.3 ms of weight streaming per FLUX
        forward).  Returns False -- and prepares nothing -- when batch > 4 or batch * nsteps > 8 (the plain path then applies)."""
        ts = torch.as_tensor(timesteps, dtype=torch.float32)
        if ts.dim() == 1:
            ts = ts[:, None].expand(-1, batch)
        nsteps, B = ts.shape
        if B > 4 or B * nsteps > 8:
            return False
        dev = self.device
        ts = ts.to(dev).contiguous()
        g = None if guidance is None else guidance.to(dev, torch.float32).expand(B).contiguous()
        pooled = None if pooled_projections is None else pooled_projections.to(dev, torch.bfloat16).contiguous()
        self._workspace(B, n_img, n_txt)
        _lib.check(self.lib.afx_mmdit_prepare_steps(self._ctx, _ptr(pooled), _ptr(ts), _ptr(g), B, nsteps, _stream()))
        self._prepared = (nsteps, B)
        return True

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
                pooled_projections: Optional[torch.Tensor] = None, guidance: Optional[torch.Tensor] = None,
                height_tokens: Optional[int] = None, width_tokens: Optional[int] = None, stage: int = 0,
                prepared_step: Optional[int] = None):
        """hidden_states [B,N,C] packed latents; timestep [B] sigma in [0,1]; encoder_hidden_states [B,T,joint];
        returns ArcFlowModelOutput (student) or the velocity [B,N,C] (teacher head).
        stage 1 / 2 (B <= 4): conditioning + embedders only / norm_out + head only -- the caller runs the blocks in between
        on ``export('x_tokens')`` and hands the result back with ``import_tokens`` (training with LoRA dropout)."""
        B, N, Cc = hidden_states.shape
        T = encoder_hidden_states.shape[1]
        if height_tokens is None:
            height_tokens = width_tokens = int(round(N ** 0.5))
        assert height_tokens * width_tokens == N, 'pass height_tokens/width_tokens for non-square latents'
        if B > 4:   # the library runs micro-batches of 4 (2 problems per sample in a grouped launch); staged calls do not
            assert stage == 0, 'staged forwards take at most 4 samples'
        dev = self.device
        x = hidden_states.to(dev, torch.bfloat16).contiguous()
        ctx = encoder_hidden_states.to(dev, torch.bfloat16).contiguous()
        t = timestep.to(dev, torch.float32).expand(B).contiguous()
        g = None if guidance is None else guidance.to(dev, torch.float32).expand(B).contiguous()
        pooled = None if pooled_projections is None else pooled_projections.to(dev, torch.bfloat16).contiguous()
        cos, sin = self.rope_tables(height_tokens, width_tokens, T)
        ws_key_before = self._ws_key
        self._workspace(min(B, 4), N, T)           # (a grown workspace drops the prepared steps: the plain path applies)
        if self._ws_key != ws_key_before:
            self._prepared = None
        ws_key_before = self._ws_key
        K, L = self.num_gaussians, self.logweights_channels
        if self.teacher_head:
            means = torch.empty(B, N, Cc, dtype=torch.bfloat16, device=dev)
            logw = logg = None
        else:
            means = torch.empty(B, N, K, Cc, dtype=torch.bfloat16, device=dev)
            logw = torch.empty(B, N, K, L, dtype=torch.bfloat16, device=dev)
            logg = torch.empty(B, N, K - 1, L, dtype=torch.bfloat16, device=dev)
        if prepared_step is not None and getattr(self, '_prepared', None) is not None and self._prepared[1] == B \
                and 0 <= prepared_step < self._prepared[0] and stage == 0 and self._ws_key == ws_key_before:
            _lib.check(self.lib.afx_mmdit_use_prepared_step(self._ctx, prepared_step))
        _lib.check(self.lib.afx_mmdit_forward_stage(self._ctx, _ptr(x), _ptr(ctx), _ptr(pooled), _ptr(t), _ptr(g), _ptr(cos),
                                                    _ptr(sin), B, N, T, _ptr(means), _ptr(logw), _ptr(logg), stage, _stream()))
        if self.teacher_head:
            return means
        return ArcFlowModelOutput(means, logw, logg)

    __call__ = forward

    def enable_fp8(self, on: bool = True) -> None:
        """fp8 linear mode (BASELINE.json configs[4] "fp8 MFMA fwd"): quantise every block linear's weight row-wise to OCP
        e4m3 (kept next to the bf16 copy) and run those GEMMs on the 2x-rate fp8 MFMA with per-token activation scales.
        Embedders, modulation, head and attention stay bf16.  A reduced-precision OPTION: never the default, never the bench."""
        from . import ops
        if on:
            names = []
            for i in range(self.num_double):
                names += [f'd{i}.{s}_{n}' for s in ('img', 'txt') for n in ('qkv', 'out', 'mlp1', 'mlp2')]
            names += [f's{i}.{n}' for i in range(self.num_single) for n in ('fused', 'out')]
            extra = {}
            for n in names:
                if n + '.weight_q' not in self._weights:
                    q, sc = ops.quant_rows_fp8(self._weights[n + '.weight'])
                    extra[n + '.weight_q'], extra[n + '.wscale'] = q, sc
            if extra:
                self.bind_packed(extra)
        _lib.check(self.lib.afx_set_fp8_linear(self._ctx, int(on)))
        self._ws, self._ws_key = None, (0, 0, 0)          # the workspace grows by the quantised-operand buffer

    def set_temb_override(self, temb_t: Optional[torch.Tensor]) -> None:
        """[B, D] fp32 replacing timestep_embedder(t) in the next forwards (None clears); the caller keeps it alive."""
        self._temb_override = temb_t
        _lib.check(self.lib.afx_set_temb_override(self._ctx, _ptr(temb_t)))

    def import_tokens(self, x_tokens: torch.Tensor, B: int, N: int, T: int) -> None:
        _lib.check(self.lib.afx_mmdit_import_tokens(self._ctx, _ptr(x_tokens), B, N, T, _stream()))
