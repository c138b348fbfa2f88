"""Checkpoint / adapter interchange (arcflow_amd/train/checkpoint.py): key layout of the reference's iter_N.pth and
adapter directory on CPU; save -> resume and export -> load_arcflow_adapter round trips on the GPU."""
import json
import os

import pytest
import torch

from arcflow_amd.train import checkpoint as CK
from arcflow_amd.train.trunk import lora_targets


def _layout(family='flux', nd=2, ns=3, D=64, K=16, C=64, L=4, rank=8):
    n_head = K * C + K * L + (K - 1) * L
    head_n = (n_head + 127) // 128 * 128
    off = [0]
    for s in (head_n * D, head_n, 2 * D * D, 2 * D):
        off.append(off[-1] + s)
    lora, o = [], off[-1]
    for name, _key, _row0, out_f, in_f in lora_targets(family, nd, ns, D):
        lora.append((name, out_f, in_f, o, o + rank * in_f))
        o += rank * in_f + out_f * rank
    return CK.TrainableLayout(K, C, L, D, head_n, off, rank, lora), o


def test_state_dict_names_follow_the_reference_layout():
    lay, n = _layout()
    flat = torch.arange(n, dtype=torch.float32)
    sd = CK.flat_to_state(flat, lay, peft_names=True)
    # heads + norm_out: the freeze_exclude list of configs/flux/arcflux_2nfe_k16.py:20-25
    assert sd['proj_out_means.weight'].shape == (1024, 64) and sd['proj_out_logweights.bias'].shape == (64,)
    assert sd['proj_out_loggamma.weight'].shape == (60, 64) and sd['norm_out.linear.weight'].shape == (128, 64)
    # LoRA: peft names of the lora_target_modules of the same config (:40-48), adapter name 'default'
    for k in ('transformer_blocks.1.ff.net.0.proj.lora_A.default.weight', 'transformer_blocks.0.ff_context.net.2.lora_B.default.weight',
              'single_transformer_blocks.2.proj_mlp.lora_A.default.weight', 'single_transformer_blocks.0.proj_out.lora_B.default.weight'):
        assert k in sd, k
    assert sd['single_transformer_blocks.0.proj_out.lora_A.default.weight'].shape == (8, 5 * 64)
    assert sd['transformer_blocks.0.ff.net.2.lora_B.default.weight'].shape == (64, 8)
    # every element of the flat buffer except the head padding rows is owned by exactly one tensor
    covered = sum(v.numel() for v in sd.values())
    pad = (lay.head_n - 1148) * (lay.D + 1)
    assert covered + pad == n
    # and the inverse mapping restores it (accepting both the peft and the exported spelling)
    back = torch.zeros(n)
    assert CK.state_to_flat({k: v.half() for k, v in sd.items()}, back, lay) == []
    exported = CK.export_adapter_state({CK.EMA_PREFIX + k: v for k, v in sd.items()}, ema=True)
    assert 'transformer_blocks.1.ff.net.0.proj.lora_A.weight' in exported and not any('default' in k for k in exported)
    back2 = torch.zeros(n)
    extra = CK.state_to_flat(dict(exported, **{'time_text_embed.guidance_embedder.linear_1.lora_A.weight': torch.zeros(8, 256)}), back2, lay)
    assert extra == ['time_text_embed.guidance_embedder.linear_1.lora_A.weight']
    assert 'time_text_embed.timestep_embedder.linear_2.lora_B.weight' in exported            # arcflux_2nfe_k16.py:46-47
    mask = torch.ones(n, dtype=torch.bool)
    hw = mask[:lay.offsets[1]].view(lay.head_n, lay.D)
    hw[1148:] = False
    mask[lay.offsets[1] + 1148:lay.offsets[2]] = False
    assert torch.equal(back2[mask], flat[mask])
    with pytest.raises(KeyError):
        CK.state_to_flat({'proj_out_means.weight': sd['proj_out_means.weight']}, back, lay)
    with pytest.raises(ValueError):
        CK.state_to_flat(dict(sd, **{'proj_out_means.bias': torch.zeros(3)}), back, lay)


def test_qwen_layout_skips_last_txt_mlp():
    lay, _ = _layout('qwen', nd=3, ns=0)
    names = [l[0] for l in lay.lora]
    assert 'transformer_blocks.2.img_mlp.net.2' in names and 'transformer_blocks.1.txt_mlp.net.0.proj' in names
    assert not any(n.startswith('transformer_blocks.2.txt_mlp') for n in names)     # arcqwen_2nfe_k16.py:52-56 range(59)


def _tiny_distiller(lora_rank=64, optimizer='adamw'):
    from arcflow_amd.train import ArcFlowDistiller, DistillConfig
    from tests.test_distill import _setup
    cfg, w = _setup()
    dc = DistillConfig(num_decay_iters=4, warmup_iters=0, grad_clip_begin_iter=0, ema_start_iter=0, lora_rank=lora_rank, optimizer=optimizer)
    return ArcFlowDistiller('flux', dict(num_double=1, num_single=1, heads=2, joint_dim=128, pooled_dim=64), w, dc), w


def _cond(B=2, hp=8, wp=8, T=12, seed=3):
    g = torch.Generator().manual_seed(seed)
    pe = (torch.randn(B, T, 128, generator=g) * 0.5).bfloat16().cuda()
    pooled = (torch.randn(B, 64, generator=g) * 0.5).bfloat16().cuda()
    return dict(prompt_embeds=pe, pooled=pooled, hp=hp, wp=wp)


@pytest.mark.gpu
def test_checkpoint_resume_is_exact(tmp_path):
    a, _ = _tiny_distiller()
    rng = torch.Generator(device='cuda').manual_seed(5)
    for _ in range(2):
        a.train_step(_cond(), 2, rng=rng)
    path = CK.save_checkpoint(a, str(tmp_path), fp16=False)
    assert os.path.basename(path) == 'iter_2.pth' and os.path.islink(tmp_path / 'latest.pth')
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert ck['meta']['iter'] == 2 and any(k.startswith('diffusion_ema.denoising.') for k in ck['state_dict'])
    b, _ = _tiny_distiller()
    meta = CK.load_checkpoint(b, str(tmp_path / 'latest.pth'))
    assert meta['iter'] == 2 and b.iteration == 2 and b.opt_steps == a.opt_steps
    lay = CK.layout_of(a)
    sa, sb = CK.flat_to_state(a.params, lay), CK.flat_to_state(b.params, lay)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(CK.flat_to_state(a.ema, lay)['norm_out.linear.weight'], CK.flat_to_state(b.ema, lay)['norm_out.linear.weight'])
    # the next step of the resumed run equals the next step of the original one
    ra, rb = (torch.Generator(device='cuda').manual_seed(9) for _ in range(2))
    ia, ib = a.train_step(_cond(seed=4), 2, rng=ra), b.train_step(_cond(seed=4), 2, rng=rb)
    assert abs(float(ia['loss']) - float(ib['loss'])) <= 1e-6 * max(1.0, abs(float(ia['loss'])))
    for k, v in CK.flat_to_state(a.params, lay).items():
        assert torch.allclose(v, CK.flat_to_state(b.params, lay)[k], rtol=0, atol=1e-7), k


@pytest.mark.gpu
def test_background_save_snapshots_the_state_at_call_time(tmp_path):
    """``save_checkpoint_async`` (tools/train.py): the file written by the thread holds the state of the moment of the call -- training goes on
    while it is written -- appears under its final name only when complete, and equals what the synchronous writer stores."""
    a, _ = _tiny_distiller()
    rng = torch.Generator(device='cuda').manual_seed(5)
    a.train_step(_cond(), 2, rng=rng)
    ref = CK.build_checkpoint(a, fp16=False)
    path = CK.save_checkpoint_async(a, str(tmp_path), fp16=False)
    a.train_step(_cond(seed=3), 2, rng=rng)               # the live state moves on while the writer runs
    CK.wait_pending_save()
    assert os.path.basename(path) == 'iter_1.pth' and os.path.exists(path) and not os.path.exists(path + '.tmp')
    assert os.readlink(tmp_path / 'latest.pth') == 'iter_1.pth'
    got = torch.load(path, map_location='cpu', weights_only=False)
    assert got['meta']['iter'] == 1 and set(got['state_dict']) == set(ref['state_dict'])
    for k, v in ref['state_dict'].items():
        assert torch.equal(got['state_dict'][k], v), k
    live = CK.build_checkpoint(a, fp16=False)['state_dict']
    assert any(not torch.equal(live[k], ref['state_dict'][k]) for k in live)          # ... and it really did move on


@pytest.mark.gpu
def test_adamw8bit_distiller_first_step_equals_fp32_and_resumes_exactly(tmp_path):
    """`optimizer='adamw8bit'` (the reference's bitsandbytes class, _ddp_train.py:18-26): block-wise 8-bit moments for the groups of
    >= 4096 values, fp32 for the small ones.  The parameter update uses the un-quantised new moments, so the FIRST step from
    zero moments equals the fp32 optimizer to the last bits; later steps differ by the quantisation of the carried moments; the 8-bit
    state survives a checkpoint round trip exactly."""
    a, _ = _tiny_distiller(optimizer='adamw8bit')
    f, _ = _tiny_distiller(optimizer='adamw')
    ra, rf = (torch.Generator(device='cuda').manual_seed(5) for _ in range(2))
    a.train_step(_cond(), 2, rng=ra)
    f.train_step(_cond(), 2, rng=rf)
    assert a.exp_avg is None and a.opt8 and all(b - a_ >= 4096 for a_, b in a.opt8) and all(b - a_ < 4096 for a_, b in a._small)
    # bitsandbytes granularity: one state per parameter TENSOR -- every LoRA A / B matrix, the three head matrices and
    # norm_out.linear.weight are their own 8-bit groups (absmax blocks start at tensor starts), every bias keeps fp32 moments
    groups = {(x, y) for x, y, _ in a.optimizer_groups()}
    assert groups == set(a.opt8) | set(a._small)
    for sp in a.trunk.specs:
        assert (sp.off_a, sp.off_a + a.trunk.r * sp.in_f) in groups and (sp.off_b, sp.off_b + sp.out_f * a.trunk.r) in groups
    assert (a._off[3], a._off[4]) in a._small and (a._off[2], a._off[3]) in a.opt8            # norm_out bias fp32, weight 8 bit
    assert sum(1 for x, y in a._small if a._off[1] <= x < a._off[2]) == 3                       # the three head biases
    assert torch.allclose(a.params, f.params, rtol=0, atol=1e-7)        # (gradients of two runs differ by ~1e-10: atomics)
    a.train_step(_cond(seed=6), 2, rng=ra)
    f.train_step(_cond(seed=6), 2, rng=rf)
    d = float((a.params - f.params).norm() / (f.params - _tiny_distiller()[0].params).norm())
    assert 0 < d < 0.1, d            # second step: same direction, a few % apart (moments carried in 8 bits)
    st_bytes = sum(s.state1.numel() + s.state2.numel() + 4 * (s.absmax1.numel() + s.absmax2.numel()) for s in a.opt8.values())
    assert st_bytes < 0.26 * 8 * sum(b - a_ for a_, b in a.opt8)                # 2 B + 2 x 4 B / 256 per value against 8 B
    path = CK.save_checkpoint(a, str(tmp_path), fp16=False)
    assert torch.load(path, map_location='cpu', weights_only=False)['optimizer']['diffusion']['format'] == 'arcflow_amd.flat_adamw8bit'
    b, _ = _tiny_distiller(optimizer='adamw8bit')
    CK.load_checkpoint(b, path)
    assert b.opt_steps == a.opt_steps == 2 and set(b.opt8) == set(a.opt8)
    for k in a.opt8:
        assert torch.equal(a.opt8[k].state1, b.opt8[k].state1) and torch.equal(a.opt8[k].absmax2, b.opt8[k].absmax2)
    r1, r2 = (torch.Generator(device='cuda').manual_seed(9) for _ in range(2))
    a.train_step(_cond(seed=4), 2, rng=r1)
    b.train_step(_cond(seed=4), 2, rng=r2)
    assert torch.allclose(a.params, b.params, rtol=0, atol=1e-7)
    # a state written for another layout (ranges that are not this distiller's tensors) is refused with a warning, not mis-applied
    ck = torch.load(path, map_location='cpu', weights_only=False)
    g8 = ck['optimizer']['diffusion']['groups8']
    k0 = sorted(g8)[0]
    g8[(k0[0] + 128, k0[1] + 128)] = g8.pop(k0)
    torch.save(ck, str(tmp_path / 'other_layout.pth'))
    c, _ = _tiny_distiller(optimizer='adamw8bit')
    with pytest.warns(UserWarning, match='different parameter layout'):
        CK.load_checkpoint(c, str(tmp_path / 'other_layout.pth'))
    assert not c.opt8 and c.opt_steps == 0


@pytest.mark.gpu
def test_exported_adapter_loads_into_the_pipeline(tmp_path):
    from arcflow_amd.pipelines import ArcFluxPipeline
    d, w = _tiny_distiller()
    rng = torch.Generator(device='cuda').manual_seed(5)
    for _ in range(2):
        d.train_step(_cond(), 2, rng=rng)
    out = CK.export_adapter(d, str(tmp_path / 'adapter'), ema=False, policy_kwargs=dict(denoising_mean_mode='U'))
    cfg = json.load(open(os.path.join(out, 'config.json')))
    assert cfg['_class_name'] == 'ArcFluxTransformer2DModel' and cfg['num_gaussians'] == 16 and cfg['logweights_channels'] == 4
    base = {k: v for k, v in w.items() if not k.startswith('proj_out_')}
    tcfg = dict(num_layers=1, num_single_layers=1, num_attention_heads=2, attention_head_dim=128, joint_attention_dim=128,
                pooled_projection_dim=64, in_channels=64, guidance_embeds=True)
    pipe = ArcFluxPipeline.from_state_dict(tcfg, base, student=False)
    assert pipe.load_arcflow_adapter(out) == 'transformer_arcflow'
    assert pipe.policy_config == {'denoising_mean_mode': 'U', 'type': 'ArcFlow'}
    # the adapted student of the pipeline (LoRA folded at load) == the distiller's live student (LoRA merged on request)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 64, 64, generator=g).bfloat16().cuda()
    c = _cond(B=1)
    t = torch.tensor([0.7], device='cuda')
    gd = torch.full((1,), 3.5, device='cuda')
    d.trunk.bind_merged()              # training runs the adapters unmerged; the engine gets W + B A on request
    o1 = d.student.forward(x, t, c['prompt_embeds'], c['pooled'], gd, 8, 8)
    o2 = pipe.transformer.forward(x, t, c['prompt_embeds'], c['pooled'], gd, 8, 8)
    for k in ('means', 'logweights', 'loggammas'):
        a, b = o1[k].float(), o2[k].float()
        assert (a - b).norm() <= 2e-2 * a.norm() + 1e-3, k
