"""CPU-side checks of the C-ABI boundary: the library builds/loads, exports every symbol that
include/arcflow_hip.h declares, validates arguments, and the host logic (schedule, rope tables,
weight packing) agrees with the oracle.  No GPU compute is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from arcflow_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'arcflow_hip.h')).read()
    names = set(re.findall(r'^(?:int|int64_t|const char\*)\s+(afx_[a-z0-9_]+)\s*\(', hdr, flags=re.M))
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f'{n} declared in arcflow_hip.h but not exported'
    from arcflow_amd import _lib
    assert set(_lib.EXPORTS) == names


def test_argument_validation_no_gpu(lib):
    from arcflow_amd import _lib
    desc = _lib.ModelDesc(0, 1, 1, 2, 64, 64, 128, 64, 1, 16, 4, 0)      # head_dim != 128
    ctx = C.c_void_p()
    assert lib.afx_create(C.byref(desc), C.byref(ctx)) == -5
    assert b'head_dim' in lib.afx_last_error()
    desc = _lib.ModelDesc(0, 1, 1, 2, 128, 64, 128, 64, 1, 16, 4, 0)
    assert lib.afx_create(C.byref(desc), C.byref(ctx)) == 0
    assert lib.afx_finalize(ctx) == -2                                    # weights missing
    assert b'not bound' in lib.afx_last_error()
    assert lib.afx_workspace_bytes(ctx, 1, 64, 16) > 0
    assert lib.afx_workspace_bytes(ctx, 0, 64, 16) == -1
    assert lib.afx_mmdit_forward(ctx, None, None, None, None, None, None, None, 1, 4, 4, None, None, None, None) == -1
    assert lib.afx_linear_bf16(None, 8, None, 8, None, None, 8, 1, 8, 64, 0, 0, None, 0, 1, None, 0, None) == -1
    assert lib.afx_arcflow_step(None, None, None, None, 1, 1.0, 1.0, 0.5, None, 1e-4, None, 1, 1, 16, 64, 4, None) == -1
    # round-5 entry points: nulls, misaligned / non-multiple-of-8 widths are refused before any launch
    assert lib.afx_linear_tn_f32out(None, 8, None, 8, None, 8, 64, 8, 8, 0, None) == -1
    assert lib.afx_linear_tn_f32out(C.c_void_p(4096), 8, C.c_void_p(8192), 8, C.c_void_p(16384), 8, 64, 12, 8, 0, None) == -1      # N1 % 8
    assert lib.afx_linear_tn_f32out(C.c_void_p(4096 + 2), 8, C.c_void_p(8192), 8, C.c_void_p(16384), 8, 64, 8, 8, 0, None) == -1   # 16-byte alignment
    assert b'afx_linear_tn_f32out' in lib.afx_last_error()
    assert lib.afx_linear_bf16_dropres(None, 64, None, 64, None, 8, 1, 8, 64, None, 8, 0.05, 1, 0, None) == -1
    assert lib.afx_linear_bf16_dropres(C.c_void_p(4096), 64, C.c_void_p(8192), 64, C.c_void_p(16384), 8, 1, 8, 64, C.c_void_p(16384), 8, 1.5, 1, 0, None) == -1   # p >= 1
    assert lib.afx_normout_backward_split(None, 8, None, 8, None, None, 1, 8, None) == -1
    # round-6: the token-split TN product: workspace sizes are a host-side function of the shape; the _ws entry refuses a missing / misaligned workspace before any launch
    if os.environ.get('AFX_TN_SPLIT') is None:
        assert lib.afx_linear_tn_ws_bytes(4608, 3072, 256) == 9 * 3072 * 256 * 4          # 48 tiles -> 9 runs of 8 K-steps
        assert lib.afx_linear_tn_ws_bytes(4608, 12288, 256) == 2 * 12288 * 256 * 4
        assert lib.afx_linear_tn_ws_bytes(130, 64, 192) == 0 and lib.afx_linear_tn_ws_bytes(1, 8, 8) == 0
        assert lib.afx_linear_tn_f32out_ws(C.c_void_p(4096), 3072, C.c_void_p(8192), 256, C.c_void_p(16384), 256, 4608, 3072, 256, 0, None, None) == -1     # needs a workspace
        assert b'workspace' in lib.afx_last_error()
    assert lib.afx_linear_tn_ws_bytes(-1, 8, 8) == -1
    assert lib.afx_linear_tn_f32out_ws(None, 8, None, 8, None, 8, 64, 8, 8, 0, None, None) == -1
    assert lib.afx_linear_tn_f32out_ws(C.c_void_p(4096), 8, C.c_void_p(8192), 8, C.c_void_p(16384), 8, 64, 8, 8, 0, C.c_void_p(4096 + 4), None) == -1   # misaligned workspace
    # round-6: the capability query behind ops.linear_dropres' fallback follows the kernel choice (host-side state only)
    if os.environ.get('AFX_GEMM_IMPL', '3') == '3' and os.environ.get('AFX_GEMM_SK', '0') == '0':
        assert lib.afx_gemm_dropres_available() == 1
        assert lib.afx_gemm_set_mode(2, 0) == 0 and lib.afx_gemm_dropres_available() == 0
        assert lib.afx_gemm_set_mode(3, 0) == 0 and lib.afx_gemm_dropres_available() == 1
    assert lib.afx_destroy(ctx) == 0


def test_schedule_matches_oracle_and_golden(golden):
    from arcflow_amd.schedule import FlowMatchEulerDiscreteScheduler, retrieve_raw_timesteps
    from oracle import arcflow_ref as R
    g = golden('g1_time_grid')
    for nfe, ratio in [(2, 1.0), (4, 1.0), (4, 0.5), (1, 1.0), (3, 0.25), (8, 1.0)]:
        tag = f'n{nfe}_r{str(ratio).replace(".", "p")}'
        raw, counts, total = retrieve_raw_timesteps(nfe, 128, ratio)
        assert np.array_equal(np.asarray(raw), g[tag + '_raw'])
        assert counts == g[tag + '_counts'].tolist() and total == int(g[tag + '_total'])
    sch = FlowMatchEulerDiscreteScheduler.from_config(dict(num_train_timesteps=1000, use_dynamic_shifting=True),
                                                      shift=3.2, shift_terminal=None, use_dynamic_shifting=False)
    raw, counts, total = retrieve_raw_timesteps(2, 128, 1.0)
    ts = sch.set_timesteps(sigmas=raw, mu=1.15)
    assert len(ts) == 128
    sig, _ = R.inference_sigmas(2)
    assert abs(float(ts[0]) / 1000 - sig[0]) < 1e-7 and abs(float(ts[64]) / 1000 - sig[1]) < 1e-6


def test_rope_tables_match_oracle():
    from arcflow_amd import rope
    from oracle import dit_ref as D
    c, s = rope.flux_tables(5, 7, 9)
    oc, os_ = D.flux_rope_tables(5, 7, 9)
    assert torch.equal(c, oc) and torch.equal(s, os_)
    c, s = rope.qwen_tables(6, 4, 5)
    ia, ta = D.qwen_rope_angles(6, 4, 5)
    assert torch.allclose(c, torch.cat([torch.cos(ta), torch.cos(ia)]), atol=1e-6)


def test_weight_packing_and_lora_merge():
    from arcflow_amd.weights import merge_lora, pack_flux
    from oracle import dit_ref as D
    cfg = D.FluxCfg(num_layers=1, num_single_layers=1, heads=2, joint_dim=128, pooled_dim=64)
    w = D.make_flux_weights(cfg, 0)
    p = pack_flux(w, 1, 1, 'cpu')
    Dm = 256
    assert p['d0.img_qkv.weight'].shape == (3 * Dm, Dm)
    assert torch.equal(p['d0.img_qkv.weight'][2 * Dm:], w['transformer_blocks.0.attn.to_q.weight'])
    assert torch.equal(p['s0.fused.weight'][:Dm], w['single_transformer_blocks.0.attn.to_k.weight'])
    assert torch.equal(p['s0.fused.weight'][3 * Dm:], w['single_transformer_blocks.0.proj_mlp.weight'])
    assert p['mod.weight'].shape == (12 * Dm + 3 * Dm + 2 * Dm, Dm)
    assert p['head.weight'].shape == (1152, Dm) and p['head.weight'][1148:].abs().sum() == 0
    g = torch.Generator().manual_seed(0)
    a = torch.randn(8, Dm, generator=g).bfloat16()
    b = torch.randn(4 * Dm, 8, generator=g).bfloat16()
    name = 'transformer_blocks.0.ff.net.0.proj'
    merged = merge_lora(w, {name + '.lora_A.weight': a, name + '.lora_B.weight': b})
    ref = (w[name + '.weight'].float() + b.float() @ a.float()).bfloat16()
    assert torch.equal(merged[name + '.weight'], ref)
    assert merged['x_embedder.weight'] is w['x_embedder.weight']


def test_phase_weights_reproduce_conv_of_nearest_upsample():
    """arcflow_amd.vae.phase_weights (host logic of afx_upconv3x3_bf16): conv3x3(nearest-2x upsample(x)) == the four 2x2 phase convolutions on
    the low-resolution input, interleaved -- in fp32 torch on the CPU (the kernel side is tests/test_vae.py::test_upsample_folded_into_conv_vs_torch)."""
    import torch
    from arcflow_amd.vae import phase_weights
    g = torch.Generator().manual_seed(0)
    co, ci, H, W = 8, 64, 5, 6
    wt = (torch.randn(co, ci, 3, 3, generator=g) * 0.1).bfloat16()
    x = torch.randn(1, ci, H, W, generator=g)
    w9 = wt.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()
    w4 = phase_weights(w9, ci).float().reshape(4, co, 2, 2, ci)
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x, scale_factor=2, mode='nearest'), wt.float(), padding=1)[0]
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))[0]                          # zero border, like the padded grid
    out = torch.zeros(co, 2 * H, 2 * W)
    for py in (0, 1):
        for px in (0, 1):
            k = w4[2 * py + px].permute(0, 3, 1, 2)                           # [co, ci, 2, 2]
            # source rows {y - 1 + py, y + py} -> padded rows {y + py, y + py + 1}
            ph = torch.nn.functional.conv2d(xp[None, :, py:py + H + 1, px:px + W + 1], k)[0]
            out[:, py::2, px::2] = ph
    # the phase kernels are sums of bf16 taps rounded once more to bf16
    assert ((out - ref).norm() / ref.norm()).item() < 5e-3
