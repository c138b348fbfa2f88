"""Host-side rotary tables for the joint [text; image] sequence (cos, sin as [S, 64] fp32).

FLUX:  FluxPosEmbed(theta=1e4, axes=(16,56,56)) on ids text=(0,0,0), image=(0,row,col)
       (reference lakonlab/models/architecture/arcflow/arcflux.py:51,171-173,360-373); the reference
       casts the tables to the trunk dtype (bf16) before use (:173), reproduced with bf16_round=True.
Qwen:  QwenEmbedRope(theta=1e4, axes=(16,56,56), scale_rope=True): centred image positions, text
       positions continue from max(h//2, w//2) on all axes (arcqwen.py:46,134).
Both rotate consecutive (even, odd) pairs, so one kernel serves both.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch


def flux_tables(hp: int, wp: int, n_txt: int, axes: Sequence[int] = (16, 56, 56), theta: float = 10000.0,
                bf16_round: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    rows = torch.arange(hp, dtype=torch.float64).repeat_interleave(wp)
    cols = torch.arange(wp, dtype=torch.float64).repeat(hp)
    pos = [torch.zeros(hp * wp, dtype=torch.float64), rows, cols]
    ang = []
    for a, d in enumerate(axes):
        omega = theta ** (-torch.arange(0, d, 2, dtype=torch.float64) / d)
        ang.append(pos[a][:, None] * omega[None, :])
    img = torch.cat(ang, dim=-1)
    full = torch.cat([torch.zeros(n_txt, img.shape[1], dtype=torch.float64), img], dim=0)
    cos, sin = torch.cos(full).float(), torch.sin(full).float()
    if bf16_round:
        cos, sin = cos.bfloat16().float(), sin.bfloat16().float()
    return cos.contiguous(), sin.contiguous()


def qwen_tables(hp: int, wp: int, n_txt: int, axes: Sequence[int] = (16, 56, 56), theta: float = 10000.0
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    om = [1.0 / torch.pow(torch.tensor(theta), torch.arange(0, d, 2, dtype=torch.float32) / d) for d in axes]

    def centred(n: int) -> torch.Tensor:
        return torch.cat([torch.arange(-(n - n // 2), 0), torch.arange(0, n // 2)]).float()
    rows = centred(hp).repeat_interleave(wp)
    cols = centred(wp).repeat(hp)
    img = torch.cat([torch.zeros(hp * wp, om[0].numel()), rows[:, None] * om[1][None, :],
                     cols[:, None] * om[2][None, :]], dim=-1)
    start = max(hp // 2, wp // 2)
    tpos = torch.arange(start, start + n_txt).float()
    txt = torch.cat([tpos[:, None] * o[None, :] for o in om], dim=-1)
    full = torch.cat([txt, img], dim=0)
    return torch.cos(full).contiguous(), torch.sin(full).contiguous()
