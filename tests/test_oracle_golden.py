"""Pin oracle/arcflow_ref.py against the golden vectors produced by the reference's own
functions (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import arcflow_ref as R


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=2e-6, atol=2e-6):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f'max abs err {err}'


@pytest.mark.parametrize('nfe,ratio', [(2, 1.0), (4, 1.0), (4, 0.5), (1, 1.0), (3, 0.25), (8, 1.0)])
def test_g1_time_grid(golden, nfe, ratio):
    g = golden('g1_time_grid')
    tag = f'n{nfe}_r{str(ratio).replace(".", "p")}'
    raw, counts, total = R.raw_time_grid(nfe, 128, ratio)
    assert counts == g[tag + '_counts'].tolist()
    assert total == int(g[tag + '_total'])
    assert np.array_equal(np.asarray(raw, dtype=np.float64), g[tag + '_raw'])   # bit exact (fp64)


def test_g1_two_nfe_sigmas():
    sig, counts = R.inference_sigmas(2, 128, 1.0, 3.2)
    assert counts == [64, 64]
    assert sig[0] == 1.0 and sig[2] == 0.0
    assert abs(sig[1] - 3.2 * 0.5 / (1 + 2.2 * 0.5)) < 1e-6       # 0.76190...


def test_g2_step_pipeline(golden):
    g = golden('g2_step_pipeline')
    means, logw, logg, x = T(g['means']), T(g['logw']), T(g['logg']), T(g['x'])
    for i in range(3):
        s_src = float(g[f'case{i}_sigma_src'])
        s_end = float(np.float32(g[f'case{i}_t_end']) / np.float32(1000))
        close(g[f'case{i}_sigma_end'].reshape(()), s_end, atol=1e-7)
        out = R.momentum_step(x, means, logw, logg, s_src, s_src, s_end, 1e-4)
        close(out, g[f'case{i}_x_end'])
        # x0-means the policy object stores (policies/arcflow.py:41-50): x - sigma * means
        close(x.unsqueeze(1) - s_src * means, g[f'case{i}_x0_means'])
    assert int(g['qwen_num_returns']) == 4
    close(R.momentum_step(x, means, logw, logg, 1.0, 1.0, float(np.float32(761.9047761) / 1000)), g['qwen_x_end'])


def test_g3_step_training(golden):
    g = golden('g3_step_training')
    means, logw, logg, x = T(g['means']), T(g['logw']), T(g['logg']), T(g['x'])
    close(R.shift_sigma(T(g['warp_in'])), g['warp_out'], atol=1e-7)
    close(R.unshift_sigma(T(g['warp_in'])), g['unwarp_out'], atol=1e-7)
    s_src = R.shift_sigma(T(g['raw_src'])).reshape(-1, 1, 1, 1)
    s_a = R.shift_sigma(T(g['raw_a'])).reshape(-1, 1, 1, 1)
    s_end = R.shift_sigma(T(g['raw_end'])).reshape(-1, 1, 1, 1)
    close(s_src, g['sigma_src'], atol=1e-7)
    close(s_end, g['sigma_end'], atol=1e-7)
    close(s_end.flatten() * 1, g['t_end'], atol=1e-7)
    close(R.momentum_step(x, means, logw, logg, s_src, s_a, s_end), g['x_end'])


def test_g4_velocity(golden):
    g = golden('g4_velocity')
    means, logw, logg, x = T(g['means']), T(g['logw']), T(g['logg']), T(g['x'])
    s_src, s_a, raw_a = T(g['sigma_src']), T(g['sigma_a']), T(g['raw_a'])
    close(R.policy_velocity(means, logw, logg, s_src, s_a), g['velocity'])
    for tag in ('', '_long', '_short'):
        pred = R.mean_velocity(x, means, logw, logg, s_src, s_a, raw_a, T(g['raw_b' + tag]))
        close(pred, g['pred' + tag], rtol=2e-5, atol=2e-5)


def test_g5_layouts(golden):
    g = golden('g5_layouts')
    hp, wp = int(g['hp']), int(g['wp'])
    lat = T(g['lat'])
    assert torch.equal(R.pack_latents(lat), T(g['packed']))
    assert torch.equal(R.unpack_latents(T(g['packed']), hp, wp), T(g['unpacked']))
    assert torch.equal(R.unpack_latents(R.pack_latents(lat), hp, wp), lat)
    m, lw, lg = R.unpack_mixture(T(g['means_tok']), T(g['logw_tok']), T(g['logg_tok']), hp, wp)
    assert torch.equal(m, T(g['means_lat']))
    assert torch.equal(lw, T(g['logw_lat']))
    assert torch.equal(lg, T(g['logg_lat']))
    # training-side twins are the same permutation: patchify == pack, unpatchify == unpack_mp
    pat = R.patchify(lat)
    assert torch.equal(pat, T(g['patchified']))
    assert torch.equal(pat.flatten(2).permute(0, 2, 1), T(g['packed']))
    tm = T(g['means_tok']).permute(0, 2, 3, 1).reshape(2, 16, 64, hp, wp)
    assert torch.equal(R.unpatchify(tm), T(g['unp_means']))
    assert torch.equal(T(g['unp_means']), T(g['means_lat']))
    assert torch.equal(T(g['unp_logw']), T(g['logw_lat']))
    assert torch.equal(T(g['unp_logg']), T(g['logg_lat']))
    # whole pipeline step carried out directly in the token layout
    out = R.momentum_step_packed(T(g['x_tok']), T(g['means_tok']), T(g['logw_tok']), T(g['logg_tok']),
                                 1.0, 1.0, float(np.float32(761.9047761) / 1000))
    close(out, g['x_end_tok'])


def _check_g9(step, inp, g, rtol, atol):
    """`step(x_tok, means_tok, logw_tok, logg_tok, sigma_src, sigma_end) -> x_end_tok` against fixture G9 (one FULL-SIZE step of the reference:
    latent [1, 16, 128, 128], both steps of the 2-NFE schedule): the strided sample element by element, the fp64 moments over all 262 144 outputs."""
    stride = int(g['stride'])
    for i in range(2):
        s_src, s_end = float(g[f'case{i}_sigma_src']), float(np.float32(g[f'case{i}_t_end']) / 1000)
        out = step(inp['x_tok'], inp['means_tok'], inp['logw_tok'], inp['logg_tok'], s_src, s_end).detach().cpu().reshape(-1)
        assert out.numel() == 4096 * 64
        ref = T(g[f'case{i}_sample'])
        err = (out[::stride].double() - ref.double()).abs().max().item()
        assert torch.allclose(out[::stride].double(), ref.double(), rtol=rtol, atol=atol), (i, err)
        n = out.numel()
        assert abs(out.double().sum().item() - float(g[f'case{i}_sum'])) <= atol * n ** 0.5 * 4, i
        ssq = float(g[f'case{i}_sumsq'])
        assert abs((out.double() ** 2).sum().item() - ssq) <= 4 * rtol * ssq + 1e-3, i
        assert abs(out.abs().max().item() - float(g[f'case{i}_absmax'])) <= 1e-4 * float(g[f'case{i}_absmax']), i


def test_g9_step_full_size(golden_full_size_inputs):
    inp, g = golden_full_size_inputs
    _check_g9(lambda x, m, lw, lg, s0, s1: R.momentum_step_packed(x, m, lw, lg, s0, s0, s1), inp, g, rtol=2e-6, atol=2e-6)


def test_g6_misc(golden):
    g = golden('g6_misc')
    for tag, p in (('01', 0.1), ('97', 0.97)):
        mask = R.gm_dropout_mask(T(g['u' + tag]), p)
        full = mask.expand_as(T(g['dropped' + tag]))
        assert torch.equal(full, T(g['dropped' + tag]))
        assert not mask.all(dim=1).any()          # never every component
    pos, neg = T(g['pos']), T(g['neg'])
    close(R.cfg_bias(pos, neg, 4.0, False), g['cfg_plain'])
    close(R.cfg_bias(pos, neg, 4.0, True), g['cfg_orth'], rtol=1e-5, atol=1e-5)
    for it, beta in zip(g['ema_steps'], g['ema_betas']):
        assert abs(R.karras_ema_beta(int(it), 100, 7.0) - beta) < 1e-15
    close(R.loggamma_bias_init(15, 4), g['loggamma_bias'], atol=1e-6)


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_g7_segment(golden, tag):
    g = golden('g7_segment_' + tag)
    means, logw, logg, x = T(g['means']), T(g['logw']), T(g['logg']), T(g['x'])

    def teacher(x_t, t):
        return 0.3 * x_t - 0.7 * t.reshape(-1, 1, 1, 1) + 0.05 * torch.roll(x_t, 1, dims=-1)

    mask = R.gm_dropout_mask(T(g['u_drop']), 0.1)
    captured = {}
    orig = R.flow_mse_loss

    def spy(p, t, scale=30.0):
        captured['p'], captured['t'] = p, t
        return orig(p, t, scale)
    R.flow_mse_loss = spy
    try:
        loss, x_dst, raw_dst = R.segment_distill(
            teacher, x, means, logw, logg, T(g['raw_src']), float(g['teacher_ratio']), float(g['segment']),
            T(g['u_student']), T(g['u_teacher']), drop_mask=mask)
    finally:
        R.flow_mse_loss = orig
    close(raw_dst, g['raw_dst'], atol=1e-7)
    close(captured['t'], g['u_tgt'], rtol=1e-5, atol=1e-5)
    close(captured['p'], g['u_pred'], rtol=1e-4, atol=1e-4)
    close(x_dst, g['x_dst'], rtol=1e-5, atol=1e-5)
    close(loss, g['loss'], rtol=1e-5, atol=1e-5)
