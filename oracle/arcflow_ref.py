"""CPU oracle for the ArcFlow-owned math on the 2-NFE hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package ``arcflow_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker.

This file is an independent restatement (torch CPU, fp32 unless a dtype is
passed) of the algorithms the reference owns.  Every function cites the
reference location it follows (paths relative to /root/reference):

* time grid ............ lakonlab/pipelines/arcflux_pipeline.py:34-70
* sigma shift .......... lakonlab/models/diffusions/sampler.py:46-48 (training twin of the
                         diffusers FlowMatchEulerDiscreteScheduler static shift,
                         inference_flux.py:14-15)
* pack / unpack ........ lakonlab/pipelines/arcflux_pipeline.py:135-193,
                         lakonlab/models/architecture/arcflow/arcflux.py:375-407
* analytic step ........ lakonlab/pipelines/arcflux_pipeline.py:195-249,
                         lakonlab/models/diffusions/arcflow.py:28-79
* policy velocity ...... lakonlab/models/diffusions/policies/arcflow.py:52-76
* mean velocity ........ lakonlab/models/diffusions/arcflow.py:81-110
* GM dropout ........... lakonlab/models/diffusions/policies/arcflow.py:96-106
* CFG bias ............. lakonlab/models/diffusions/gaussian_flow.py:18-26
* segment distillation . lakonlab/models/diffusions/arcflow.py:120-209
* MSE flow loss ........ lakonlab/models/losses/diffusion_loss.py:44-83
* Karras EMA ........... lakonlab/runner/hooks/ema_hook.py:86-89
* log-gamma head init .. lakonlab/models/architecture/arcflow/arcflux.py:103-132

Parity status: PINNED.  ``tests/golden/make_golden.py`` executes the reference's
own functions (in the build container, never on the GPU box) and commits their
inputs/outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
this file against every one of them.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------- time grid
def raw_time_grid(nfe: int, total_substeps: int = 128, timestep_ratio: float = 1.0
                  ) -> Tuple[List[float], List[int], int]:
    """Sub-step grid of the policy integrator (arcflux_pipeline.py:34-70).

    Every NFE covers a segment of raw time ``1/(nfe-1+ratio)`` (the last one is
    scaled by ``ratio``); each segment is cut into ``round(seg*total)`` (>=1)
    equally spaced sub-steps starting at the segment's upper end.
    """
    seg = 1.0 / (nfe - 1 + timestep_ratio)
    grid: List[float] = []
    counts: List[int] = []
    t_hi = 1.0
    for i in range(nfe):
        width = seg if i < nfe - 1 else seg * timestep_ratio
        n = max(round(width * total_substeps), 1)
        counts.append(n)
        # np.linspace(endpoint=False) == t_hi + j * (-(width)/n) evaluated in float64
        pts = np.linspace(t_hi, t_hi - width, n, endpoint=False)
        grid.extend(np.clip(pts, 0.0, None).tolist())
        t_hi = t_hi - width
    return grid, counts, sum(counts)


def shift_sigma(t, shift: float = 3.2):
    """Static resolution shift sigma = s t / (1 + (s-1) t)   (sampler.py:46-48)."""
    return shift * t / (1 + (shift - 1) * t)


def unshift_sigma(sigma, shift: float = 3.2):
    """Inverse of :func:`shift_sigma` (sampler.py:50-52)."""
    return sigma / (shift + (1 - shift) * sigma)


def inference_sigmas(nfe: int, total_substeps: int = 128, timestep_ratio: float = 1.0,
                     shift: float = 3.2) -> Tuple[List[float], List[int]]:
    """sigma at the start of every NFE plus the terminal 0 (arcflux_pipeline.py:414-493).

    The scheduler stores float32 sigmas, so the shift is evaluated in float32 like
    FlowMatchEulerDiscreteScheduler.set_timesteps(sigmas=...) does.
    """
    grid, counts, total = raw_time_grid(nfe, total_substeps, timestep_ratio)
    sig = np.asarray(grid, dtype=np.float32)
    sig = (np.float32(shift) * sig / (np.float32(1) + np.float32(shift - 1) * sig)).astype(np.float32)
    out, idx = [], 0
    for n in counts:
        out.append(float(sig[idx]))
        idx += n
    out.append(0.0)
    return out, counts


# --------------------------------------------------------------------------- layouts
def pack_latents(x: Tensor, patch: int = 2) -> Tensor:
    """[B,C,H,W] -> [B,(H/p)(W/p), C*p*p], channel index = c*p*p + ph*p + pw
    (arcflux_pipeline.py:162-175 with patch_size=1,target_patch_size=2)."""
    b, c, h, w = x.shape
    hp, wp = h // patch, w // patch
    t = x.reshape(b, c, hp, patch, wp, patch)
    return t.permute(0, 2, 4, 1, 3, 5).reshape(b, hp * wp, c * patch * patch)


def unpack_latents(tok: Tensor, hp: int, wp: int, patch: int = 2) -> Tensor:
    """Inverse of :func:`pack_latents` (arcflux_pipeline.py:177-193, target_patch_size=1)."""
    b, n, ch = tok.shape
    c = ch // (patch * patch)
    t = tok.reshape(b, hp, wp, c, patch, patch)
    return t.permute(0, 3, 1, 4, 2, 5).reshape(b, c, hp * patch, wp * patch)


def unpack_mixture(means: Tensor, logw: Tensor, logg: Tensor, hp: int, wp: int, patch: int = 2
                   ) -> Tuple[Tensor, Tensor, Tensor]:
    """Token-major head outputs -> latent-space mixture (arcflux_pipeline.py:135-160).

    means [B,N,K,C*p*p] -> [B,K,C,H,W];  logw [B,N,K,p*p] -> [B,K,1,H,W];
    logg [B,N,K-1,p*p] -> [B,K-1,1,H,W].
    """
    def spread(t: Tensor) -> Tensor:
        b, n, k, ch = t.shape
        c = ch // (patch * patch)
        t = t.reshape(b, hp, wp, k, c, patch, patch)
        return t.permute(0, 3, 4, 1, 5, 2, 6).reshape(b, k, c, hp * patch, wp * patch)
    return spread(means), spread(logw), spread(logg)


def patchify(x: Tensor, patch: int = 2) -> Tensor:
    """Training-side fold [B,C,H,W] -> [B,C*p*p,H/p,W/p] (arcflux.py:375-384)."""
    b, c, h, w = x.shape
    t = x.reshape(b, c, h // patch, patch, w // patch, patch)
    return t.permute(0, 1, 3, 5, 2, 4).reshape(b, c * patch * patch, h // patch, w // patch)


def unpatchify(t: Tensor, patch: int = 2) -> Tensor:
    """Training-side unfold [B,K,C*p*p,h,w] -> [B,K,C,h*p,w*p] (arcflux.py:386-407)."""
    b, k, ch, h, w = t.shape
    c = ch // (patch * patch)
    t = t.reshape(b, k, c, patch, patch, h, w)
    return t.permute(0, 1, 2, 5, 3, 6, 4).reshape(b, k, c, h * patch, w * patch)


# --------------------------------------------------------------------------- analytic step
def _phi(z: Tensor, eps: float) -> Tensor:
    """phi(z) = expm1(z)/z with |z| clamped to >= eps keeping the sign (0 -> +eps)
    (arcflux_pipeline.py:227-233)."""
    sgn = torch.where(z < 0, -torch.ones_like(z), torch.ones_like(z))
    zs = sgn * z.abs().clamp(min=eps)
    return torch.expm1(zs) / zs


def momentum_step(x: Tensor, means: Tensor, logw: Tensor, logg: Tensor,
                  sigma_src, sigma_start, sigma_end, eps: float = 1e-4) -> Tensor:
    """Closed-form ArcFlow transport of x from sigma_start to sigma_end.

    x [B,C,H,W]; means [B,K,C,H,W]; logw [B,K,1,H,W]; logg [B,K-1,1,H,W];
    sigmas are python floats or tensors broadcastable to [B,1,1,1].

        x_end = x - D * sum_k softmax(logw)_k m_k d_k phi_k,
        d_0 = phi_0 = 1, d_k = exp(g_k (s_src - s_start)), phi_k = phi(g_k D), D = s_start - s_end

    (arcflux_pipeline.py:205-247, arcflow.py:46-77).
    """
    def as4(s):
        s = torch.as_tensor(s, dtype=x.dtype)
        return s.reshape(-1, 1, 1, 1) if s.dim() > 0 else s.reshape(1, 1, 1, 1)
    s_src, s_a, s_b = as4(sigma_src), as4(sigma_start), as4(sigma_end)
    d_past = (s_src - s_a).unsqueeze(1)      # [B,1,1,1,1]
    d_step = (s_a - s_b).unsqueeze(1)
    w = torch.softmax(logw, dim=1)
    decay = torch.exp(logg * d_past)          # [B,K-1,1,H,W]
    phi = _phi(logg * d_step, eps)
    lin = w[:, :1] * means[:, :1]                                   # k = 0 : straight line
    cur = w[:, 1:] * (means[:, 1:] * decay * d_step * phi)          # k >= 1 : exponential velocity
    disp = (lin * d_step).sum(dim=1) + cur.sum(dim=1)
    return x - disp


def momentum_step_packed(x_tok: Tensor, means_tok: Tensor, logw_tok: Tensor, logg_tok: Tensor,
                         sigma_src: float, sigma_start: float, sigma_end: float,
                         eps: float = 1e-4, patch: int = 2) -> Tensor:
    """Same step, directly in the transformer's token layout (what the HIP kernel does).

    x_tok [B,N,C*p*p]; means_tok [B,N,K,C*p*p]; logw_tok [B,N,K,p*p]; logg_tok [B,N,K-1,p*p].
    Channel ch = c*p*p + q uses the (log-)weights of sub-pixel q = ch % (p*p).
    Equivalent to unpack -> momentum_step -> pack (arcflux_pipeline.py:482-510).
    """
    pp = patch * patch
    b, n, k, ch = means_tok.shape
    x = x_tok.to(torch.float32)
    m = means_tok.to(torch.float32)
    lw = logw_tok.to(torch.float32)
    lg = logg_tok.to(torch.float32)
    q = torch.arange(ch, device=x.device) % pp
    lw_c = lw[..., q]                       # [B,N,K,ch]
    lg_c = lg[..., q]                       # [B,N,K-1,ch]
    w = torch.softmax(lw_c, dim=2)
    d_past = sigma_src - sigma_start
    d_step = sigma_start - sigma_end
    decay = torch.exp(lg_c * d_past)
    phi = _phi(lg_c * d_step, eps)
    disp = w[:, :, 0] * m[:, :, 0] * d_step + (w[:, :, 1:] * (m[:, :, 1:] * decay * d_step * phi)).sum(dim=2)
    return x - disp


def policy_velocity(means: Tensor, logw: Tensor, logg: Tensor, sigma_src, sigma_t) -> Tensor:
    """u(sigma_t) = sum_k w_k m_k d_k  (policies/arcflow.py:52-76)."""
    s_src = torch.as_tensor(sigma_src, dtype=means.dtype)
    s_t = torch.as_tensor(sigma_t, dtype=means.dtype)
    s_src = s_src.reshape(-1, 1, 1, 1) if s_src.dim() > 0 else s_src.reshape(1, 1, 1, 1)
    s_t = s_t.reshape(-1, 1, 1, 1) if s_t.dim() > 0 else s_t.reshape(1, 1, 1, 1)
    d_past = (s_src - s_t).unsqueeze(1)
    w = torch.softmax(logw, dim=1)
    decay = torch.exp(logg * d_past)
    return (w[:, :1] * means[:, :1]).sum(dim=1) + (w[:, 1:] * means[:, 1:] * decay).sum(dim=1)


def mean_velocity(x_a: Tensor, means: Tensor, logw: Tensor, logg: Tensor,
                  sigma_src: Tensor, sigma_a: Tensor, raw_a: Tensor, raw_b: Tensor,
                  total_substeps: int = 128, shift: float = 3.2, eps: float = 1e-4) -> Tensor:
    """Average velocity over [raw_b, raw_a] used by the distillation loss (arcflow.py:81-110).

    Roll-outs shorter than two sub-steps use the local velocity instead (numerically stable).
    sigma_src / sigma_a are [B,1,1,1]; raw_a / raw_b are [B].
    """
    b = x_a.shape[0]
    short = torch.round((raw_a - raw_b) * total_substeps) < 2
    sigma_b = shift_sigma(raw_b, shift).reshape(b, 1, 1, 1)
    x_b = momentum_step(x_a, means, logw, logg, sigma_src, sigma_a, sigma_b, eps)
    u_mean = (x_a - x_b) / (sigma_a - sigma_b).clamp(min=eps)
    u_loc = policy_velocity(means, logw, logg, sigma_src, sigma_a)
    return torch.where(short.reshape(b, 1, 1, 1), u_loc, u_mean)


def gm_dropout_mask(u: Tensor, p: float) -> Tensor:
    """Mask of dropped mixture components from uniforms u [B,K,1,1,1]
    (policies/arcflow.py:96-106): drop where u < p unless that drops every component."""
    drop = u < p
    all_dropped = drop.all(dim=1, keepdim=True)
    return drop & ~all_dropped


def cfg_bias(pos: Tensor, neg: Tensor, scale: float, orthogonal: bool = False) -> Tensor:
    """Classifier-free-guidance bias (gaussian_flow.py:18-26); teacher u = pos + bias."""
    bias = (pos - neg) * (scale - 1)
    if orthogonal:
        dims = list(range(1, pos.dim()))
        proj = (bias * pos).mean(dim=dims, keepdim=True) / (pos * pos).mean(dim=dims, keepdim=True).clamp(min=1e-6)
        bias = bias - proj * pos
    return bias


def flow_mse_loss(u_pred: Tensor, u_tgt: Tensor, scale: float = 30.0) -> Tensor:
    """DiffusionMSELoss with the constant rescale (diffusion_loss.py:44-83 + mmgen DDPMLoss
    'mean' reduction): mean_b( scale * 0.5 * mean_chw (u_pred-u_tgt)^2 )."""
    per = ((u_pred - u_tgt) ** 2).flatten(1).mean(dim=1) * 0.5
    return (per * scale).mean()


def karras_ema_beta(iteration: int, start_iter: int = 100, gamma: float = 7.0,
                    max_momentum: float = 1.0) -> float:
    """EMA momentum after training iteration ``iteration`` (0-based):
    t = max(iter + 1 - start, 1); beta = min((1 - 1/t)^(gamma+1), max) (ema_hook.py:86-89)."""
    t = max(iteration + 1 - start_iter, 1)
    return min((1 - 1 / t) ** (gamma + 1), max_momentum)


def loggamma_bias_init(num_gammas: int = 15, channels: int = 4,
                       lo: float = 0.2, hi: float = 4.0) -> Tensor:
    """Initial bias of proj_out_loggamma: log of log-spaced rates, repeated over the
    sub-pixel channels (arcflux.py:103-132)."""
    g = torch.logspace(math.log10(lo), math.log10(hi), num_gammas, base=10)
    lg = torch.log(g)
    return lg.unsqueeze(1).repeat(1, channels).flatten() if channels > 1 else lg


# --------------------------------------------------------------------------- segment distillation
def segment_intervals(u_student: Tensor, u_teacher: Tensor, teacher_ratio: float,
                      segment: float, window: float) -> Tuple[Tensor, Tensor]:
    """Scheduled trajectory mixing intervals from uniforms (arcflow.py:147-161).

    u_student [B,n], u_teacher [B,n-1] are U(0,1) draws.
    """
    b = u_student.shape[0]
    span = segment - window
    s = torch.sort(u_student * ((1 - teacher_ratio) * span), dim=-1)[0]
    s = torch.diff(s, dim=-1, prepend=torch.zeros(b, 1))
    t = torch.sort(u_teacher, dim=-1)[0]
    t = torch.diff(t, dim=-1, prepend=torch.zeros(b, 1), append=torch.ones(b, 1)) * (teacher_ratio * span)
    return s, t


def segment_distill(teacher, x_src: Tensor, means: Tensor, logw: Tensor, logg: Tensor,
                    raw_src: Tensor, teacher_ratio: float, segment: float,
                    u_student: Tensor, u_teacher: Tensor, drop_mask: Optional[Tensor] = None,
                    total_substeps: int = 128, window_substeps: int = 3, shift: float = 3.2,
                    eps: float = 1e-4, loss_scale: float = 30.0):
    """One student segment of trajectory-matching distillation (arcflow.py:120-209).

    ``teacher(x, t)`` returns the teacher velocity; the policy used for roll-outs is the
    detached mixture with ``drop_mask`` components removed (log-weight -> -inf), the one
    used for the predicted velocity is the full mixture.  Returns (loss, x_dst, raw_dst).
    """
    b = x_src.shape[0]
    n_states = u_student.shape[1]
    seg = torch.tensor([segment], dtype=torch.float32)
    n_sub = (seg * total_substeps).round().to(torch.long).clamp(min=1)
    window = torch.minimum(window_substeps * (seg / n_sub), seg)
    raw_dst = raw_src - seg
    sigma_src = shift_sigma(raw_src, shift).reshape(b, 1, 1, 1)
    logw_roll = logw if drop_mask is None else logw.masked_fill(drop_mask, float('-inf'))

    span = seg - window
    s_iv = torch.sort(u_student * ((1 - teacher_ratio) * span.unsqueeze(-1)), dim=-1)[0]
    s_iv = torch.diff(s_iv, dim=-1, prepend=torch.zeros(b, 1))
    t_iv = torch.sort(u_teacher, dim=-1)[0]
    t_iv = torch.diff(t_iv, dim=-1, prepend=torch.zeros(b, 1), append=torch.ones(b, 1)) \
        * (teacher_ratio * span.unsqueeze(-1))

    x, raw, sigma = x_src, raw_src, sigma_src
    preds, tgts = [], []
    for i in range(n_states):
        raw_a = (raw - s_iv[:, i]).clamp(min=0)
        raw_b = (raw_a - t_iv[:, i]).clamp(min=0)
        sigma_a = shift_sigma(raw_a, shift).reshape(b, 1, 1, 1)
        x_a = momentum_step(x, means, logw_roll, logg, sigma_src, sigma, sigma_a, eps)
        tgt = teacher(x_a, sigma_a.flatten())
        tgts.append(tgt)
        preds.append(mean_velocity(x_a, means, logw, logg, sigma_src, sigma_a, raw_a,
                                   raw_b - window, total_substeps, shift, eps))
        sigma_b = shift_sigma(raw_b, shift).reshape(b, 1, 1, 1)
        x = x_a + tgt * (sigma_b - sigma_a)
        raw, sigma = raw_b, sigma_b
    loss = flow_mse_loss(torch.cat(preds), torch.cat(tgts), loss_scale)
    sigma_dst = shift_sigma(raw_dst, shift).reshape(b, 1, 1, 1)
    x_dst = momentum_step(x, means, logw_roll, logg, sigma_src, sigma, sigma_dst, eps)
    return loss, x_dst, raw_dst
