"""torch-tensor wrappers over the exported building-block kernels of libarcflow_hip.so.

Used by the pipelines (analytic step), the distillation loop and the per-kernel parity tests.
No fallback: every function launches a HIP kernel or raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _cuda(t: torch.Tensor, dtype) -> torch.Tensor:
    if t.device.type != 'cuda':
        raise _lib.ArcflowHipError('arcflow_amd.ops works on GPU tensors only (no CPU fallback)')
    return t.to(dtype).contiguous()


def linear(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: str = 'none',
           gelu_col0: int = 0, gate: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           rows_per_batch: int = 0, out: Optional[torch.Tensor] = None, pre: Optional[torch.Tensor] = None,
           sk_ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epi(a @ w.T + bias + pre); a [M,K] (last-dim contiguous, may be a strided view), w [N,K] bf16.
    epilogue: 'none' | 'gelu' (tanh, on columns >= gelu_col0) | 'gate_res' (residual + gate[b] * (.)).
    pre [M,N] bf16 is added before the activation / gate (LoRA-dropout correction).
    sk_ws: a ``stream_k_workspace()`` buffer -> the launch may split its under-filled last round stream-K style."""
    lib = _lib.load()
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.stride(-1) == 1 and w.stride(-1) == 1
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    epi = {'none': 0, 'gelu': 1, 'gate_res': 2}[epilogue]
    if gate is not None:
        gate = _cuda(gate, torch.float32)
        if gate.dim() == 1:
            gate = gate[None]
    rpb = rows_per_batch if rows_per_batch > 0 else max(M, 1)
    if sk_ws is not None:
        assert pre is None
        _lib.check(lib.afx_linear_bf16_sk(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K,
                                          epi, gelu_col0, _p(gate), 0 if gate is None else gate.stride(0), rpb,
                                          _p(residual), 0 if residual is None else residual.stride(0), _p(sk_ws), _s()))
        return out
    _lib.check(lib.afx_linear_bf16_pre(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K,
                                       epi, gelu_col0, _p(gate), 0 if gate is None else gate.stride(0), rpb,
                                       _p(residual), 0 if residual is None else residual.stride(0),
                                       _p(pre), 0 if pre is None else pre.stride(0), _s()))
    return out


def set_gemm_mode(impl: int = 3, tile: int = 0) -> None:
    """Kernel / tile-shape override of every bf16 GEMM (``afx_gemm_set_mode``): impl 3 = one-wave-per-SIMD kernel with tile 0 = picked
    per launch, 1 ... 6 = 256x256 / 288x192 / 320x192 / 128x128 / 256x224 / 224x256; impl 2 = 8-phase 256x256 kernel; impl 1 = simple reference kernel."""
    _lib.check(_lib.load().afx_gemm_set_mode(impl, tile))


def set_attn_impl(impl: int = 0) -> None:
    """Kernel choice of the joint attention (``afx_attn_set_impl``): 0 = one-wave-per-SIMD kernel where eligible, the blocks of an under-filled
    last round KV-split (default), 1 = 4-wave kernel always, 2 = 8-wave ping-pong kernel (experimental), 3 = as 0 on the plain grid."""
    _lib.check(_lib.load().afx_attn_set_impl(impl))


def set_attn_bwd_impl(impl: int = 3) -> None:
    """Kernel generation of ``attention_bwd`` (``afx_attn_bwd_set_impl``): 3 = generated dK / dV + dQ streams in one launch (default), 4 = as two launches,
    1 = generated dK / dV + round-4 dQ, 2 = the round-4 kernels."""
    _lib.check(_lib.load().afx_attn_bwd_set_impl(impl))


def stream_k_workspace(device='cuda') -> torch.Tensor:
    """Zero-initialised workspace for ``linear(..., sk_ws=)`` (hand-off flags + fp32 accumulator slabs of the stream-K tail)."""
    lib = _lib.load()
    return torch.zeros(lib.afx_linear_sk_ws_bytes(), dtype=torch.uint8, device=device)


def lora_dropout(src: torch.Tensor, p: float, seed: int, row0: int = 0, mode: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Counter-based LoRA input dropout (keep probability 1-p, delta = keep/(1-p) - 1):
    mode 0: out = src * delta;  1: out = src * (1 + delta) = dropout(src);  2: out += src * delta;  3: out += src * (1 + delta)."""
    lib = _lib.load()
    M, N = src.shape
    if out is None:
        assert mode < 2
        out = torch.empty(M, N, dtype=torch.bfloat16, device=src.device)
    _lib.check(lib.afx_lora_dropout_bf16(_p(src), src.stride(0), _p(out), out.stride(0), M, N, row0, float(p), int(seed) & 0xffffffff, mode, _s()))
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """q,k,v [B,S,H,128] bf16 (last two dims contiguous) -> [B,S,H*128]; softmax(q k^T/sqrt(128)) v."""
    lib = _lib.load()
    B, S, H, Dh = q.shape
    assert Dh == 128 and q.dtype == torch.bfloat16
    q2, k2, v2 = (t.reshape(B * S, H * Dh) for t in (q.contiguous(), k.contiguous(), v.contiguous()))
    o = torch.empty(B * S, H * Dh, dtype=torch.bfloat16, device=q.device)
    ws = torch.empty(lib.afx_attention_ws_bytes(B, H, S), dtype=torch.uint8, device=q.device)
    _lib.check(lib.afx_attention_bf16(_p(q2), H * Dh, _p(k2), H * Dh, _p(v2), H * Dh, _p(o), H * Dh, _p(ws), B, H, S, _s()))
    return o.reshape(B, S, H * Dh)


def attention_to_mx8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor):
    """``attention`` with the output as a block-scaled fp8 operand: (uint8 [B*S, H*128] e4m3, uint8 [B*S, H] E8M0 -- one scale per token and head)."""
    lib = _lib.load()
    B, S, H, Dh = q.shape
    assert Dh == 128 and q.dtype == torch.bfloat16
    q2, k2, v2 = (t.reshape(B * S, H * Dh) for t in (q.contiguous(), k.contiguous(), v.contiguous()))
    o8 = torch.empty(B * S, H * Dh, dtype=torch.uint8, device=q.device)
    mx = torch.empty(B * S, (H + 3) // 4 * 4, dtype=torch.uint8, device=q.device)
    ws = torch.empty(lib.afx_attention_ws_bytes(B, H, S), dtype=torch.uint8, device=q.device)
    _lib.check(lib.afx_attention_to_mx8(_p(q2), H * Dh, _p(k2), H * Dh, _p(v2), H * Dh, _p(o8), H * Dh, _p(mx), mx.stride(0), _p(ws), B, H, S, _s()))
    return o8, mx[:, :H]


def norm_modulate(x: torch.Tensor, scale: torch.Tensor, shift: Optional[torch.Tensor], rows_per_batch: int = 0,
                  rms: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [R,D] bf16; AdaLN: scale/shift [B,D] f32 (row r uses batch r // rows_per_batch);
    rms=True: x * rsqrt(mean x^2 + 1e-6) * scale[D]."""
    lib = _lib.load()
    R, D = x.shape
    if out is None:
        out = torch.empty(R, D, dtype=torch.bfloat16, device=x.device)
    scale = _cuda(scale, torch.float32)
    shift = None if shift is None else _cuda(shift, torch.float32)
    ldm = 0 if rms or scale.dim() == 1 else scale.stride(0)
    _lib.check(lib.afx_norm_modulate_bf16(_p(x), x.stride(0), _p(out), out.stride(0), R, D, _p(scale), _p(shift), ldm,
                                          rows_per_batch if rows_per_batch > 0 else max(R, 1), int(rms), _s()))
    return out


def qk_norm_rope_(x: torch.Tensor, w_txt: torch.Tensor, w_img: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                  n_txt: int) -> torch.Tensor:
    """In place on x [B,S,H,128] bf16: per-head RMSNorm (rows < n_txt use w_txt) then pair RoPE."""
    lib = _lib.load()
    B, S, H, Dh = x.shape
    assert x.is_contiguous() and Dh == 128
    _lib.check(lib.afx_qk_norm_rope_bf16(_p(x), H * Dh, _p(_cuda(w_txt, torch.float32)), _p(_cuda(w_img, torch.float32)),
                                         _p(_cuda(cos, torch.float32)), _p(_cuda(sin, torch.float32)), B, S, n_txt, H, _s()))
    return x


def gemv(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: str = 'none',
         out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    lib = _lib.load()
    x = _cuda(x, torch.float32)
    B, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.zeros(B, N, dtype=torch.float32, device=x.device)
    _lib.check(lib.afx_gemv_bf16(_p(x), _p(w), _p(bias), _p(out), B, N, K, 1 if act == 'silu' else 0, int(accumulate), _s()))
    return out


def _mix_dtype(means, logw, logg) -> Tuple[int, torch.Tensor, torch.Tensor, torch.Tensor]:
    if means.dtype == torch.bfloat16:
        return _lib.AFX_DT_BF16, means.contiguous(), logw.to(torch.bfloat16).contiguous(), logg.to(torch.bfloat16).contiguous()
    return _lib.AFX_DT_F32, means.float().contiguous(), logw.float().contiguous(), logg.float().contiguous()


def arcflow_step(x: torch.Tensor, means: torch.Tensor, logw: torch.Tensor, logg: torch.Tensor, sigma_src, sigma_start,
                 sigma_end, eps: float = 1e-4, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Analytic ArcFlow transport in the token layout: x [B,N,ch] f32, means [B,N,K,ch], logw [B,N,K,pp],
    logg [B,N,K-1,pp]  ->  x_end [B,N,ch] f32.  sigmas: python floats, or [B] tensors (per-sample)."""
    lib = _lib.load()
    B, N, K, ch = means.shape
    pp = logw.shape[-1]
    x = _cuda(x, torch.float32)
    dt, means, logw, logg = _mix_dtype(means, logw, logg)
    if out is None:
        out = torch.empty_like(x)
    sv = None
    if any(isinstance(s, torch.Tensor) and s.numel() > 1 for s in (sigma_src, sigma_start, sigma_end)):
        cols = [torch.as_tensor(s, dtype=torch.float32, device=x.device).flatten().expand(B) for s in (sigma_src, sigma_start, sigma_end)]
        sv = torch.stack(cols, dim=1).contiguous()
        s0 = s1 = s2 = 0.0
    else:
        s0, s1, s2 = (float(s) for s in (sigma_src, sigma_start, sigma_end))
    _lib.check(lib.afx_arcflow_step(_p(x), _p(means), _p(logw), _p(logg), dt, s0, s1, s2, _p(sv), eps, _p(out),
                                    B, N, K, ch, pp, _s()))
    return out


def arcflow_velocity(means: torch.Tensor, logw: torch.Tensor, logg: torch.Tensor, sigma_src, sigma_t) -> torch.Tensor:
    """u(sigma_t) of the momentum mixture, token layout -> [B,N,ch] f32."""
    lib = _lib.load()
    B, N, K, ch = means.shape
    pp = logw.shape[-1]
    dt, means, logw, logg = _mix_dtype(means, logw, logg)
    out = torch.empty(B, N, ch, dtype=torch.float32, device=means.device)
    sv = None
    if any(isinstance(s, torch.Tensor) and s.numel() > 1 for s in (sigma_src, sigma_t)):
        cols = [torch.as_tensor(s, dtype=torch.float32, device=out.device).flatten().expand(B) for s in (sigma_src, sigma_t, sigma_t)]
        sv = torch.stack(cols, dim=1).contiguous()
        s0 = s1 = 0.0
    else:
        s0, s1 = float(sigma_src), float(sigma_t)
    _lib.check(lib.afx_arcflow_velocity(_p(means), _p(logw), _p(logg), dt, s0, s1, _p(sv), _p(out), B, N, K, ch, pp, _s()))
    return out


# ---------------------------------------------------------------------------------------------------
# distillation-step kernels
def _sigma_vec(B, device, *sig):
    cols = [torch.as_tensor(x, dtype=torch.float32, device=device).flatten().expand(B) for x in sig]
    return torch.stack(cols, dim=1).contiguous()


def arcflow_step_dropout(x, means, logw, logg, sigma_src, sigma_start, sigma_end, drop_mask=None, eps: float = 1e-4):
    """Roll-out step of the detached policy with per-sample sigmas and GM dropout mask [B,K] (bool/uint8)."""
    lib = _lib.load()
    B, N, K, ch = means.shape
    pp = logw.shape[-1]
    x = _cuda(x, torch.float32)
    dt, means, logw, logg = _mix_dtype(means, logw, logg)
    sv = _sigma_vec(B, x.device, sigma_src, sigma_start, sigma_end)
    dm = None if drop_mask is None else drop_mask.to(device=x.device, dtype=torch.uint8).reshape(B, K).contiguous()
    out = torch.empty_like(x)
    _lib.check(lib.afx_arcflow_step_dropout(_p(x), _p(means), _p(logw), _p(logg), dt, _p(sv), _p(dm), eps, _p(out),
                                            B, N, K, ch, pp, _s()))
    return out


def arcflow_backward(g, means, logw, logg, sigma_src, sigma_start, sigma_end, gscale=None, velocity: bool = False,
                     grads=None, eps: float = 1e-4):
    """Gradients of the displacement (or velocity) w.r.t. the mixture.  g [B,N,ch] upstream gradient, gscale [B]
    optional per-sample factor.  grads = (d_means, d_logw, d_logg) fp32 to accumulate into, else fresh tensors."""
    lib = _lib.load()
    B, N, K, ch = means.shape
    pp = logw.shape[-1]
    g = _cuda(g, torch.float32)
    dt, means, logw, logg = _mix_dtype(means, logw, logg)
    sv = _sigma_vec(B, g.device, sigma_src, sigma_start, sigma_end)
    gs = None if gscale is None else _cuda(torch.as_tensor(gscale, device=g.device).flatten().expand(B), torch.float32)
    acc = grads is not None
    if not acc:
        grads = (torch.empty(B, N, K, ch, dtype=torch.float32, device=g.device),
                 torch.empty(B, N, K, pp, dtype=torch.float32, device=g.device),
                 torch.empty(B, N, K - 1, pp, dtype=torch.float32, device=g.device))
    _lib.check(lib.afx_arcflow_backward(_p(g), _p(means), _p(logw), _p(logg), dt, 0.0, 0.0, 0.0, _p(sv), _p(gs), 1.0, eps,
                                        _p(grads[0]), _p(grads[1]), _p(grads[2]), B, N, K, ch, pp, int(velocity), int(acc), _s()))
    return grads


def mse_loss(pred, target, coef: float, loss_accum: torch.Tensor, want_grad: bool = True):
    lib = _lib.load()
    pred, target = _cuda(pred, torch.float32), _cuda(target, torch.float32)
    grad = torch.empty_like(pred) if want_grad else None
    _lib.check(lib.afx_mse_loss(_p(pred), _p(target), coef, _p(grad), _p(loss_accum), pred.numel(), _s()))
    return grad


def euler_roll(x_a, u, sigma_a, sigma_b):
    lib = _lib.load()
    x_a, u = _cuda(x_a, torch.float32), _cuda(u, torch.float32)
    B = x_a.shape[0]
    sa = _cuda(sigma_a.flatten().expand(B), torch.float32)
    sb = _cuda(sigma_b.flatten().expand(B), torch.float32)
    out = torch.empty_like(x_a)
    _lib.check(lib.afx_euler_roll(_p(x_a), _p(u), _p(sa), _p(sb), _p(out), B, x_a[0].numel(), _s()))
    return out


def axpby_rows(a, alpha, b, beta):
    """alpha[s] * a + beta[s] * b with per-sample scalars alpha, beta [B]."""
    lib = _lib.load()
    a, b = _cuda(a, torch.float32), _cuda(b, torch.float32)
    B = a.shape[0]
    al, be = _cuda(alpha.flatten().expand(B), torch.float32), _cuda(beta.flatten().expand(B), torch.float32)
    out = torch.empty_like(a)
    _lib.check(lib.afx_axpby_rows(_p(a), _p(al), _p(b), _p(be), _p(out), B, a[0].numel(), _s()))
    return out


def cfg_combine(pos, neg, scale: float):
    lib = _lib.load()
    pos, neg = _cuda(pos, torch.float32), _cuda(neg, torch.float32)
    out = torch.empty_like(pos)
    _lib.check(lib.afx_cfg_combine(_p(pos), _p(neg), scale, _p(out), pos.numel(), _s()))
    return out


def head_grad(d_means, d_logw, d_logg, logw_out, ldy: int):
    lib = _lib.load()
    B, N, K, ch = d_means.shape
    lw = d_logw.shape[-1]
    dy = torch.empty(B * N, ldy, dtype=torch.bfloat16, device=d_means.device)
    _lib.check(lib.afx_head_grad(_p(d_means), _p(d_logw), _p(d_logg), _p(logw_out.contiguous()), _p(dy), ldy, B * N, K, ch, lw, _s()))
    return dy


def linear_f32out(a, w, out=None, accumulate: bool = False):
    """out (fp32) [M,N] (+)= a [M,K] @ w[N,K].T, bf16 operands."""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.zeros(M, N, dtype=torch.float32, device=a.device)
    _lib.check(lib.afx_linear_bf16_f32out(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), M, N, K,
                                          int(accumulate), _s()))
    return out


def linear_dropres(a, w, residual, p: float, seed: int, row0: int = 0, out=None):
    """out = residual + (a @ w.T) * keep/(1-p) with the LoRA-dropout mask of ``lora_dropout`` (seed, row0 + row, column) applied to the fp32 product in
    the GEMM's epilogue (one launch instead of product + mask-and-add pass).  out may be ``residual``."""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    if not lib.afx_gemm_dropres_available():
        # the masked residual add lives in the one-wave-per-SIMD kernel's epilogue only (set_gemm_mode(1 / 2), AFX_GEMM_IMPL, AFX_GEMM_SK select
        # other kernels in A/B runs and parity tests): the product to memory, then the mask-and-add pass -- the same bits
        if out.data_ptr() != residual.data_ptr():
            out.copy_(residual)
        return lora_dropout(linear(a, w), p, seed, row0, mode=3, out=out)
    _lib.check(lib.afx_linear_bf16_dropres(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), M, N, K, _p(residual), residual.stride(0),
                                           float(p), seed & 0xffffffff, row0, _s()))
    return out


def linear_tn_f32out(x, y, out=None, accumulate: bool = False):
    """out (fp32) [N1,N2] (+)= x[M,N1].T @ y[M,N2]: the contraction runs over the ROWS of two token-major bf16 matrices (row strides = their
    stride(0)); the LoRA weight gradients without transposed copies (afx_tn.hip)."""
    lib = _lib.load()
    M, N1 = x.shape
    N2 = y.shape[1]
    assert y.shape[0] == M and x.stride(1) == 1 and y.stride(1) == 1
    if out is None:
        out = torch.zeros(N1, N2, dtype=torch.float32, device=x.device)
    nws = lib.afx_linear_tn_ws_bytes(M, N1, N2)       # > 0: the token loop is cut into runs (fp32 partial tiles in ws, added in a fixed order: deterministic)
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws > 0 else None      # (from the current stream's pool: private until the launches have run)
    _lib.check(lib.afx_linear_tn_f32out_ws(_p(x), x.stride(0), _p(y), y.stride(0), _p(out), out.stride(0), M, N1, N2, int(accumulate), _p(ws), _s()))
    return out


def quant_rows_fp8(x: torch.Tensor):
    """bf16 [M,K] -> (q uint8 [M,K] OCP e4m3, scale fp32 [M]) with q = round(x / scale[r]), scale = absmax(row) / 448."""
    lib = _lib.load()
    M, K = x.shape
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    scale = torch.empty(M, dtype=torch.float32, device=x.device)
    _lib.check(lib.afx_quant_rows_fp8(_p(x), x.stride(0), _p(q), K, _p(scale), M, K, _s()))
    return q, scale


def linear_fp8(aq, a_scale, wq, w_scale, bias=None, epilogue: str = 'none', gelu_col0: int = 0, gate=None, residual=None,
               rows_per_batch: int = 0, out=None):
    """bf16 out = epi(a_scale[m] w_scale[n] (aq @ wq.T) + bias) on the 2x-rate fp8 MFMA; aq [M,K], wq [N,K] uint8 (e4m3)."""
    lib = _lib.load()
    M, K = aq.shape
    N = wq.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=aq.device)
    epi = {'none': 0, 'gelu': 1, 'gate_res': 2}[epilogue]
    if gate is not None:
        gate = _cuda(gate, torch.float32)
        if gate.dim() == 1:
            gate = gate[None]
    rpb = rows_per_batch if rows_per_batch > 0 else max(M, 1)
    _lib.check(lib.afx_linear_fp8(_p(aq), aq.stride(0), _p(a_scale), _p(wq), wq.stride(0), _p(w_scale), _p(bias), _p(out), out.stride(0),
                                  M, N, K, epi, gelu_col0, _p(gate), 0 if gate is None else gate.stride(0), rpb, _p(residual),
                                  0 if residual is None else residual.stride(0), _s()))
    return out


def quant_rows_mx8(x: torch.Tensor):
    """bf16 [M,K] -> (q uint8 [M,K] OCP e4m3, mx uint8 [M, K/128] E8M0): block-scaled fp8, one power-of-two scale per row and 128 columns
    (q = round(x * 2^(127 - mx)), the smallest scale that keeps the block inside +-448)."""
    lib = _lib.load()
    M, K = x.shape
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    nb = K // 128
    mx = torch.empty(M, (nb + 3) // 4 * 4, dtype=torch.uint8, device=x.device)
    _lib.check(lib.afx_quant_rows_mx8(_p(x), x.stride(0), _p(q), K, _p(mx), mx.stride(0), M, K, _s()))
    return q, mx[:, :nb]


def linear_fp8_mx(aq, a_mx, wq, w_scale, bias=None, epilogue: str = 'none', gelu_col0: int = 0, gate=None, residual=None,
                  rows_per_batch: int = 0, out=None, a_scale=None):
    """bf16 out = epi(w_scale[n] (sum over K-tiles t of 2^(a_mx[m, t] - 127) aq[m, t] . wq[n, t]) + bias): the fp8 GEMM on block-scaled
    activations (``quant_rows_mx8`` / the fused producers).  K % 512 == 0."""
    lib = _lib.load()
    M, K = aq.shape
    N = wq.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=aq.device)
    if a_scale is None:
        a_scale = torch.ones(M, dtype=torch.float32, device=aq.device)
    epi = {'none': 0, 'gelu': 1, 'gate_res': 2}[epilogue]
    if gate is not None:
        gate = _cuda(gate, torch.float32)
        if gate.dim() == 1:
            gate = gate[None]
    rpb = rows_per_batch if rows_per_batch > 0 else max(M, 1)
    _lib.check(lib.afx_linear_fp8_mx(_p(aq), aq.stride(0), _p(a_mx), a_mx.stride(0), _p(a_scale), _p(wq), wq.stride(0), _p(w_scale), _p(bias),
                                     _p(out), out.stride(0), M, N, K, epi, gelu_col0, _p(gate), 0 if gate is None else gate.stride(0), rpb,
                                     _p(residual), 0 if residual is None else residual.stride(0), _s()))
    return out


def linear_fp8_to_mx8(aq, a_mx, wq, w_scale, bias=None, gelu: bool = False, c8_col0: int = 0, a_scale=None):
    """The fp8 GEMM whose epilogue writes the next GEMM's block-scaled operand: returns (bf16 [M, c8_col0] or None, q uint8 [M, N - c8_col0],
    mx uint8 [M, (N - c8_col0) / 128]).  a_mx None: per-row ``a_scale`` only."""
    lib = _lib.load()
    M, K = aq.shape
    N = wq.shape[0]
    n8 = N - c8_col0
    nb = (n8 + 127) // 128
    head = torch.empty(M, c8_col0, dtype=torch.bfloat16, device=aq.device) if c8_col0 > 0 else None
    q = torch.empty(M, n8, dtype=torch.uint8, device=aq.device)
    mx = torch.empty(M, (nb + 3) // 4 * 4, dtype=torch.uint8, device=aq.device)
    if a_scale is None:
        a_scale = torch.ones(M, dtype=torch.float32, device=aq.device)
    _lib.check(lib.afx_linear_fp8_to_mx8(_p(aq), aq.stride(0), _p(a_mx), 0 if a_mx is None else a_mx.stride(0), _p(a_scale), _p(wq), wq.stride(0),
                                         _p(w_scale), _p(bias), _p(head), c8_col0 if head is not None else 8, _p(q), n8, _p(mx), mx.stride(0),
                                         c8_col0, M, N, K, int(gelu), _s()))
    return head, q, mx[:, :nb]


def linear_splitk(a, w, bias=None, residual=None, out=None, split_k: int = 0):
    """bf16 out = a @ w.T (+ bias) (+ residual) for few-row operands: split-K GEMM into per-chunk fp32 partial slabs, then one
    summing / converting pass.  Fills the chip when M x N alone gives only a handful of 256x256 tiles."""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    n = lib.afx_linear_splitk_chunks(M, N, K, split_k)
    part = torch.empty(n, M, N, dtype=torch.float32, device=a.device)
    _lib.check(lib.afx_linear_bf16_splitk(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(part), M, N, K, split_k, _s()))
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    _lib.check(lib.afx_finish_f32_bf16(_p(part), n, _p(residual), 0 if residual is None else residual.stride(0), _p(out), out.stride(0),
                                       M, N, _s()))
    return out


def transpose(x, pad_to: int = 1):
    """[R,C] bf16 (row-strided view allowed) -> [C, roundup(R, pad_to)], zero padded."""
    lib = _lib.load()
    R, Cc = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    y = (torch.zeros if Rp != R else torch.empty)(Cc, Rp, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.afx_transpose_bf16(_p(x), x.stride(0), _p(y), Rp, R, Cc, _s()))
    return y


def coldot(a, b, out_accum):
    """out_accum[c] += sum_r a[r,c] * b[r,c]   (a, b bf16 row-strided views, out fp32 [C])."""
    lib = _lib.load()
    _lib.check(lib.afx_coldot_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(out_accum), a.shape[0], a.shape[1], _s()))
    return out_accum


def gate_residual(y, gate, res, out=None):
    """out = res + gate[c] * y   (y, res bf16 [R,C], gate fp32 [C])."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(y.shape[0], y.shape[1], dtype=torch.bfloat16, device=y.device)
    _lib.check(lib.afx_gate_residual_bf16(_p(y), y.stride(0), _p(_cuda(gate, torch.float32)), _p(res), res.stride(0), _p(out), out.stride(0),
                                          y.shape[0], y.shape[1], _s()))
    return out


def gemv_t(x, w, out_accum):
    """out_accum[b,k] += sum_n x[b,n] w[n,k]   (x fp32 [B<=4, N], w bf16 [N, K], out fp32 [B, K])."""
    lib = _lib.load()
    _lib.check(lib.afx_gemv_t_bf16(_p(x), x.stride(0), _p(w), w.stride(0), _p(out_accum), x.shape[0], x.shape[1], w.shape[1], _s()))
    return out_accum


def colsum(x, out_accum):
    lib = _lib.load()
    _lib.check(lib.afx_colsum_bf16(_p(x), x.stride(0), _p(out_accum), x.shape[0], x.shape[1], _s()))
    return out_accum


def normout_backward(x, dxn, dmod_accum, rows_per_batch: int):
    lib = _lib.load()
    _lib.check(lib.afx_normout_backward(_p(x), x.stride(0), _p(dxn), dxn.stride(0), _p(dmod_accum), x.shape[0], x.shape[1],
                                        rows_per_batch, _s()))
    return dmod_accum


def normout_backward_split(x, dxn, d_scale_accum, d_shift_accum):
    """d_scale_accum[D] += sum_rows dxn * LN(x), d_shift_accum[D] += sum_rows dxn (all rows one batch entry; fp32 vectors, e.g. slices of a larger buffer)."""
    lib = _lib.load()
    assert d_scale_accum.is_contiguous() and d_shift_accum.is_contiguous() and d_scale_accum.dtype == torch.float32
    _lib.check(lib.afx_normout_backward_split(_p(x), x.stride(0), _p(dxn), dxn.stride(0), _p(d_scale_accum), _p(d_shift_accum), x.shape[0], x.shape[1], _s()))


def outer_accum(dmod, x, dW_accum):
    lib = _lib.load()
    B, J = dmod.shape
    _lib.check(lib.afx_outer_accum(_p(_cuda(dmod, torch.float32)), _p(_cuda(x, torch.float32)), _p(dW_accum), B, J, x.shape[1], _s()))
    return dW_accum


def sumsq(x, out_accum):
    lib = _lib.load()
    _lib.check(lib.afx_sumsq(_p(x), _p(out_accum), x.numel(), _s()))
    return out_accum


def adamw_step(param, grad, exp_avg, exp_avg_sq, lr, step, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    lib = _lib.load()
    _lib.check(lib.afx_adamw_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), lr, betas[0], betas[1], eps,
                                  weight_decay, step, grad_scale, param.numel(), _s()))


def dynamic_map(signed: bool = True, max_exponent_bits: int = 7, total_bits: int = 8) -> torch.Tensor:
    """The 256-entry "dynamic" 8-bit code book of bitsandbytes' block-wise optimizers (functional.create_dynamic_map; bitsandbytes
    is an unvendored, unpinned dependency of the reference -- requirements.txt:11 -- so this restates its published construction):
    for every exponent i the interval [0.1, 1] * 10^(i - 6) is cut into 2^i (signed) or 2^(i+1) (unsigned) equal cells whose
    midpoints are codes (mirrored when signed), plus 0 and 1; sorted ascending.  Signed for exp_avg, unsigned for exp_avg_sq."""
    data = []
    non_sign_bits = total_bits - 1
    additional_items = 2 ** (non_sign_bits - max_exponent_bits) - 1
    i = 0
    for i in range(max_exponent_bits):
        fraction_items = int(2 ** (i + non_sign_bits - max_exponent_bits) + 1 if signed else 2 ** (i + non_sign_bits - max_exponent_bits + 1) + 1)
        boundaries = torch.linspace(0.1, 1, fraction_items, dtype=torch.float64)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
        if signed:
            data += (-(10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
    if additional_items > 0:
        boundaries = torch.linspace(0.1, 1, additional_items + 1, dtype=torch.float64)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
        if signed:
            data += (-(10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
    data += [0.0, 1.0]
    assert len(data) == 2 ** total_bits, len(data)
    return torch.tensor(sorted(data), dtype=torch.float32)


class AdamW8bitState:
    """Block-wise 8-bit moments of ``n`` values: code bytes + one absmax per block of 256 (zero absmax = zero moments)."""
    BLOCK = 256

    def __init__(self, n: int, device):
        nb = (n + self.BLOCK - 1) // self.BLOCK
        self.n = n
        self.state1 = torch.zeros(n, dtype=torch.uint8, device=device)
        self.state2 = torch.zeros(n, dtype=torch.uint8, device=device)
        self.absmax1 = torch.zeros(nb, dtype=torch.float32, device=device)
        self.absmax2 = torch.zeros(nb, dtype=torch.float32, device=device)
        self.qmap1 = dynamic_map(True).to(device)
        self.qmap2 = dynamic_map(False).to(device)

    def moments(self):
        """Dequantised (exp_avg, exp_avg_sq) -- for tests and checkpoint conversion."""
        rep = lambda a: a.repeat_interleave(self.BLOCK)[:self.n]   # noqa: E731
        return self.qmap1[self.state1.long()] * rep(self.absmax1), self.qmap2[self.state2.long()] * rep(self.absmax2)

    def state_dict(self):
        return {k: getattr(self, k).detach().cpu() for k in ('state1', 'state2', 'absmax1', 'absmax2')}

    def load_state_dict(self, sd):
        for k in ('state1', 'state2', 'absmax1', 'absmax2'):
            getattr(self, k).copy_(sd[k])


def adamw8bit_step(param, grad, st: AdamW8bitState, lr, step, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    lib = _lib.load()
    assert param.numel() == st.n
    _lib.check(lib.afx_adamw8bit_step(_p(param), _p(grad), _p(st.state1), _p(st.state2), _p(st.absmax1), _p(st.absmax2), _p(st.qmap1),
                                      _p(st.qmap2), lr, betas[0], betas[1], eps, weight_decay, step, grad_scale, param.numel(), _s()))


def ema_lerp(ema, net, beta: float):
    lib = _lib.load()
    _lib.check(lib.afx_ema_lerp(_p(ema), _p(net), beta, ema.numel(), _s()))


def cast_bf16(x, out=None):
    lib = _lib.load()
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.afx_cast_f32_bf16(_p(x), _p(out), x.numel(), _s()))
    return out


# ---------------------------------------------------------------------------------------------------
# attention with gradient
def attention_fwd_lse(q, k, v):
    """q,k,v [B,S,H,128] bf16 -> (o [B,S,H*128] bf16, lse [B,H,S_pad] f32)."""
    lib = _lib.load()
    B, S, H, Dh = q.shape
    q2, k2, v2 = (t.reshape(B * S, H * Dh) for t in (q.contiguous(), k.contiguous(), v.contiguous()))
    o = torch.empty(B * S, H * Dh, dtype=torch.bfloat16, device=q.device)
    S_pad = (S + 63) // 64 * 64
    lse = torch.full((B, H, S_pad), float('inf'), dtype=torch.float32, device=q.device)
    ws = torch.empty(lib.afx_attention_ws_bytes(B, H, S), dtype=torch.uint8, device=q.device)
    _lib.check(lib.afx_attention_fwd_lse_bf16(_p(q2), H * Dh, _p(k2), H * Dh, _p(v2), H * Dh, _p(o), H * Dh, _p(lse), _p(ws),
                                              B, H, S, _s()))
    return o.reshape(B, S, H * Dh), lse


def attention_bwd(q, k, v, o, dout, lse):
    """-> (dq, dk, dv) each [B,S,H,128] bf16."""
    lib = _lib.load()
    B, S, H, Dh = q.shape
    flat = lambda t: t.contiguous().reshape(B * S, H * Dh)   # noqa: E731
    q2, k2, v2, o2, do2 = flat(q), flat(k), flat(v), flat(o), flat(dout)
    dq, dk, dv = (torch.empty(B * S, H * Dh, dtype=torch.bfloat16, device=q.device) for _ in range(3))
    ws = torch.empty(lib.afx_attention_bwd_ws_bytes(B, H, S), dtype=torch.uint8, device=q.device)
    ld = H * Dh
    _lib.check(lib.afx_attention_bwd_bf16(_p(q2), ld, _p(k2), ld, _p(v2), ld, _p(o2), ld, _p(do2), ld, _p(lse), _p(dq), ld,
                                          _p(dk), ld, _p(dv), ld, _p(ws), B, H, S, _s()))
    return tuple(t.reshape(B, S, H, Dh) for t in (dq, dk, dv))


# ---------------------------------------------------------------------------------------------------
# element-wise trunk backward
def ln_modulate_backward(x, dxn, scale, rows_per_batch: int = 0, dres=None, out=None):
    """dx = dres + LN^T(dxn * (1 + scale[b])); x, dxn [R,D] bf16 (row-strided views allowed), scale [B,D] f32."""
    lib = _lib.load()
    R, D = x.shape
    if out is None:
        out = torch.empty(R, D, dtype=torch.bfloat16, device=x.device)
    scale = _cuda(scale, torch.float32)
    if scale.dim() == 1:
        scale = scale[None]
    _lib.check(lib.afx_ln_modulate_backward(_p(x), x.stride(0), _p(dxn), dxn.stride(0), _p(scale), scale.stride(0),
                                            rows_per_batch if rows_per_batch > 0 else max(R, 1), _p(dres),
                                            0 if dres is None else dres.stride(0), _p(out), out.stride(0), R, D, _s()))
    return out


def qk_norm_rope(x, w_txt, w_img, cos, sin, n_txt: int, out=None, dy=None):
    """Out-of-place per-head RMSNorm + RoPE on x [B,S,H*128 view with row stride]; with dy: the backward (dx)."""
    lib = _lib.load()
    B, S, HD = x.shape
    H = HD // 128
    x2 = x.reshape(B * S, HD) if x.is_contiguous() else x.view(B * S, HD)
    if out is None:
        out = torch.empty(B * S, HD, dtype=torch.bfloat16, device=x.device)
    dy2 = None if dy is None else dy.view(B * S, HD)
    _lib.check(lib.afx_qk_norm_rope_oop_bf16(_p(x2), x2.stride(0), _p(out), out.stride(0), _p(dy2),
                                             0 if dy2 is None else dy2.stride(0), _p(_cuda(w_txt, torch.float32)),
                                             _p(_cuda(w_img, torch.float32)), _p(cos), _p(sin), B, S, n_txt, H,
                                             int(dy is not None), _s()))
    return out.view(B, S, HD)


def gelu(pre, dh=None, out=None):
    """h = gelu_tanh(pre) or, with dh, dpre = dh * gelu'(pre); [R,C] bf16 row-strided views."""
    lib = _lib.load()
    R, Cc = pre.shape
    if out is None:
        out = torch.empty(R, Cc, dtype=torch.bfloat16, device=pre.device)
    _lib.check(lib.afx_gelu_bf16(_p(pre), pre.stride(0), _p(dh), 0 if dh is None else dh.stride(0), _p(out), out.stride(0), R, Cc, _s()))
    return out


def add_scale(a, b=None, gate=None, rows_per_batch: int = 0, out=None):
    """out = (a (+ b)) * gate[batch]; a, b [R,C] bf16 row-strided, gate [B,C] f32."""
    lib = _lib.load()
    R, Cc = a.shape
    if out is None:
        out = torch.empty(R, Cc, dtype=torch.bfloat16, device=a.device)
    if gate is not None:
        gate = _cuda(gate, torch.float32)
        if gate.dim() == 1:
            gate = gate[None]
    _lib.check(lib.afx_add_scale_bf16(_p(a), a.stride(0), _p(b), 0 if b is None else b.stride(0), _p(gate),
                                      0 if gate is None else gate.stride(0), rows_per_batch if rows_per_batch > 0 else max(R, 1),
                                      _p(out), out.stride(0), R, Cc, _s()))
    return out


def attention_fwd_lse_2d(q, k, v, o, B: int, S: int, H: int):
    """Strided form: q, k, v, o are [B*S, H*128] views (row stride = their stride(0)); returns lse [B,H,S_pad]."""
    lib = _lib.load()
    S_pad = (S + 63) // 64 * 64
    lse = torch.full((B, H, S_pad), float('inf'), dtype=torch.float32, device=q.device)
    ws = torch.empty(lib.afx_attention_ws_bytes(B, H, S), dtype=torch.uint8, device=q.device)
    _lib.check(lib.afx_attention_fwd_lse_bf16(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
                                              _p(lse), _p(ws), B, H, S, _s()))
    return lse


def attention_bwd_2d(q, k, v, o, dout, lse, dq, dk, dv, B: int, S: int, H: int):
    lib = _lib.load()
    ws = torch.empty(lib.afx_attention_bwd_ws_bytes(B, H, S), dtype=torch.uint8, device=q.device)
    _lib.check(lib.afx_attention_bwd_bf16(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
                                          _p(dout), dout.stride(0), _p(lse), _p(dq), dq.stride(0), _p(dk), dk.stride(0),
                                          _p(dv), dv.stride(0), _p(ws), B, H, S, _s()))
