"""Time grid of the ArcFlow sampler and the flow-matching scheduler object the reference's entry
scripts configure (inference_flux.py:14-15: ``FlowMatchEulerDiscreteScheduler.from_config(
pipe.scheduler.config, shift=3.2, shift_terminal=None, use_dynamic_shifting=False)``).

Host-side float arithmetic only (128 numbers per image) -- nothing here runs on the GPU.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch


def retrieve_raw_timesteps(num_inference_steps: int, total_substeps: int, timestep_ratio: float
                           ) -> Tuple[List[float], List[int], int]:
    """Raw sub-step times, sub-steps per inference step and their total
    (same contract as lakonlab/pipelines/arcflux_pipeline.py:34-70)."""
    if num_inference_steps < 1:
        raise ValueError('num_inference_steps must be >= 1')
    seg = 1.0 / (num_inference_steps - 1 + timestep_ratio)
    times: List[float] = []
    per_step: List[int] = []
    upper = 1.0
    for i in range(num_inference_steps):
        size = seg * timestep_ratio if i == num_inference_steps - 1 else seg
        n = max(round(size * total_substeps), 1)
        per_step.append(n)
        times += np.linspace(upper, upper - size, n, endpoint=False).clip(min=0.0).tolist()
        upper -= size
    return times, per_step, sum(per_step)


class _Config(dict):
    """dict with attribute access, like diffusers' FrozenDict configs."""
    __getattr__ = dict.get

    def get(self, k, default=None):          # noqa: D401 - keep dict.get semantics
        return super().get(k, default)


class FlowMatchEulerDiscreteScheduler:
    """The subset of diffusers' scheduler the ArcFlow pipelines touch: config, from_config(),
    set_begin_index(), set_timesteps(sigmas=..., mu=...), .timesteps / .sigmas.

    sigma' = shift * s / (1 + (shift-1) s) for the static shift; with use_dynamic_shifting the
    exponential time shift exp(mu) / (exp(mu) + (1/s - 1)) of the stock FLUX pipeline.
    """
    _DEFAULTS = dict(num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=False, base_shift=0.5,
                     max_shift=1.15, base_image_seq_len=256, max_image_seq_len=4096, shift_terminal=None)

    def __init__(self, **kwargs: Any):
        cfg = dict(self._DEFAULTS)
        cfg.update(kwargs)
        self.config = _Config(cfg)
        self.timesteps: Optional[torch.Tensor] = None
        self.sigmas: Optional[torch.Tensor] = None
        self._begin_index = None

    @classmethod
    def from_config(cls, config: Optional[Dict[str, Any]] = None, **overrides: Any):
        cfg = dict(config or {})
        cfg.update(overrides)
        cfg = {k: v for k, v in cfg.items() if not k.startswith('_')}
        return cls(**cfg)

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None,
                      sigmas: Optional[List[float]] = None, mu: Optional[float] = None):
        if sigmas is None:
            n = num_inference_steps
            sigmas = np.linspace(1.0, 1.0 / self.config.num_train_timesteps, n)
        s = np.asarray(sigmas, dtype=np.float32)
        if self.config.use_dynamic_shifting:
            if mu is None:
                raise ValueError('mu is required with use_dynamic_shifting')
            s = (math.exp(mu) / (math.exp(mu) + (1.0 / s - 1.0))).astype(np.float32)
        else:
            sh = np.float32(self.config.shift)
            s = (sh * s / (np.float32(1) + (sh - np.float32(1)) * s)).astype(np.float32)
        if self.config.shift_terminal:
            one_minus = 1 - s
            s = (1 - one_minus / (one_minus[-1] / (1 - self.config.shift_terminal))).astype(np.float32)
        sig = torch.from_numpy(s)
        self.timesteps = (sig * self.config.num_train_timesteps).to(device)
        self.sigmas = torch.cat([sig, torch.zeros(1)]).to(device)
        return self.timesteps


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * m + (base_shift - m * base_seq_len)
