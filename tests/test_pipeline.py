"""Drop-in surface: adapter loader contract (CPU) and the 2-NFE pipeline end to end against the oracle (GPU)."""
import json
import os
import warnings

import pytest
import torch


def _write_adapter(tmp, cls_name, sd, meta=True):
    from safetensors.torch import save_file
    d = os.path.join(tmp, 'arcflow-flux-2steps')
    os.makedirs(d, exist_ok=True)
    json.dump({'_class_name': cls_name, 'num_gaussians': 16, 'logweights_channels': 4}, open(os.path.join(d, 'config.json'), 'w'))
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, 'diffusion_pytorch_model.safetensors'),
              metadata={'policy_config': json.dumps({'type': 'ArcFlow'})} if meta else None)
    return tmp


def _tiny():
    from oracle import dit_ref as D
    cfg = D.FluxCfg(num_layers=1, num_single_layers=1, heads=2, joint_dim=128, pooled_dim=64)
    w = D.make_flux_weights(cfg, seed=9, teacher_head=True)
    tcfg = dict(num_layers=1, num_single_layers=1, num_attention_heads=2, attention_head_dim=128, in_channels=64,
                joint_attention_dim=128, pooled_projection_dim=64, guidance_embeds=True)
    return cfg, w, tcfg


def test_loader_contract_cpu(tmp_path):
    from arcflow_amd.pipelines import ArcFluxPipeline
    cfg, w, tcfg = _tiny()
    pipe = ArcFluxPipeline()
    pipe._transformer_config, pipe._base_state_dict = tcfg, w
    heads = {k: v for k, v in w.items() if k.startswith('proj_out_')}
    root = _write_adapter(str(tmp_path / 'a'), 'SomethingElse', heads)
    with pytest.raises(ValueError, match="Can't find a model linked to SomethingElse"):
        pipe.load_arcflow_adapter(root, subfolder='arcflow-flux-2steps')
    root = _write_adapter(str(tmp_path / 'b'), 'ArcFluxTransformer2DModel', heads)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        assert pipe.load_arcflow_adapter(root, subfolder='arcflow-flux-2steps', target_module_name='transformer') is None
    assert any('No LoRA weights' in str(r.message) for r in rec)
    with pytest.raises(TypeError):
        pipe.load_arcflow_adapter(root, subfolder='arcflow-flux-2steps', bogus=1)
    with pytest.raises(EnvironmentError):
        pipe.load_arcflow_adapter('ymyy307/ArcFlow', subfolder='arcflow-flux-2steps')     # no network: local dirs only
    with pytest.raises(AssertionError):
        ArcFluxPipeline(policy_type='GMFlow')


@pytest.mark.gpu
def test_flux_pipeline_two_nfe_vs_oracle(tmp_path):
    from arcflow_amd import FlowMatchEulerDiscreteScheduler
    from arcflow_amd.pipelines import ArcFluxPipeline
    from oracle import arcflow_ref as R
    from oracle import dit_ref as D
    cfg, w, tcfg = _tiny()
    g = torch.Generator().manual_seed(2)
    name = 'single_transformer_blocks.0.proj_mlp'
    lora = {name + '.lora_A.weight': (torch.randn(8, 256, generator=g) * 0.05).bfloat16(),
            name + '.lora_B.weight': (torch.randn(1024, 8, generator=g) * 0.05).bfloat16()}
    adapter = {k: v for k, v in w.items() if k.startswith('proj_out_') or k.startswith('norm_out')}
    adapter['norm_out.linear.bias'] = adapter['norm_out.linear.bias'] + 0.125      # overlay must win over the base
    adapter.update(lora)
    root = _write_adapter(str(tmp_path), 'ArcFluxTransformer2DModel', adapter)

    base = {k: v for k, v in w.items() if not k.startswith('proj_out_')}
    pipe = ArcFluxPipeline.from_state_dict(tcfg, base, student=False)                # plain FLUX (teacher head)
    name_ret = pipe.load_arcflow_adapter(root, subfolder='arcflow-flux-2steps', target_module_name='transformer')
    assert name_ret == 'transformer_arcflow'
    pipe.scheduler = FlowMatchEulerDiscreteScheduler.from_config(pipe.scheduler.config, shift=3.2, shift_terminal=None,
                                                                 use_dynamic_shifting=False)
    pipe = pipe.to('cuda')
    T = 12
    pe = (torch.randn(1, T, 128, generator=g) * 0.5).bfloat16()
    pp = (torch.randn(1, 64, generator=g) * 0.5).bfloat16()
    gen = torch.Generator(device='cuda').manual_seed(42)
    out = pipe(prompt_embeds=pe, pooled_prompt_embeds=pp, num_images_per_prompt=1, width=128, height=128,
               num_inference_steps=2, generator=gen, timestep_ratio=1.0, output_type='latent').images
    assert out.shape == (1, 64, 64) and out.dtype == torch.float32

    # oracle: same noise, LoRA merged the same way, fp32 math
    gen = torch.Generator(device='cuda').manual_seed(42)
    noise = torch.randn(1, 16, 16, 16, generator=gen, device='cuda').cpu()
    x = R.pack_latents(noise)
    wm = dict(w)
    wm.update({k: v for k, v in adapter.items() if 'lora' not in k})
    wm[name + '.weight'] = (w[name + '.weight'].float() + lora[name + '.lora_B.weight'].float() @ lora[name + '.lora_A.weight'].float()).bfloat16()
    sig, _ = R.inference_sigmas(2)
    for i in range(2):
        m, lw, lg = D.flux_forward(wm, cfg, x.bfloat16().float(), pe.float(), pp.float(), torch.tensor([sig[i]]),
                                   torch.tensor([3.5]), 8, 8)
        x = R.momentum_step_packed(x, m, lw, lg, sig[i], sig[i], sig[i + 1])
    err = ((out.cpu() - x).norm() / x.norm()).item()
    assert err < 2.5e-2, err

    # runtime LoRA scale (arcflux.py:147-154 scale_lora_layers): y = W x + s B A x for this call, scale 1 again on the next one
    def ref_latents(scale):
        wr = dict(wm)
        wr[name + '.weight'] = (w[name + '.weight'].float() + scale * (lora[name + '.lora_B.weight'].float() @ lora[name + '.lora_A.weight'].float())).bfloat16()
        xr = R.pack_latents(noise)
        for i in range(2):
            m, lw, lg = D.flux_forward(wr, cfg, xr.bfloat16().float(), pe.float(), pp.float(), torch.tensor([sig[i]]), torch.tensor([3.5]), 8, 8)
            xr = R.momentum_step_packed(xr, m, lw, lg, sig[i], sig[i], sig[i + 1])
        return xr

    def run(**kw):
        return pipe(prompt_embeds=pe, pooled_prompt_embeds=pp, width=128, height=128, num_inference_steps=2, timestep_ratio=1.0,
                    generator=torch.Generator(device='cuda').manual_seed(42), output_type='latent', **kw).images.cpu()
    half = run(joint_attention_kwargs={'scale': 0.5})
    x_half = ref_latents(0.5)
    assert ((half - x_half).norm() / x_half.norm()).item() < 2.5e-2
    assert ((half - out.cpu()).norm() / x.norm()).item() > 1e-3                   # the scale really changed the network
    assert torch.equal(run(), out.cpu())                                           # un-scaled again (the fold is redone and cached)
    pipe.set_adapters(name_ret, adapter_weights=0.0)                               # weight 0: the base trunk + the adapter's heads / norm_out
    x_zero = ref_latents(0.0)
    assert ((run() - x_zero).norm() / x_zero.norm()).item() < 2.5e-2
    pipe.set_adapters([name_ret])
    assert torch.equal(run(), out.cpu())
    with pytest.raises(ValueError, match='not loaded'):
        pipe.set_adapters('style_lora')

    # nfe=4 / ratio 0.5 default path also runs and the callback sees every step
    seen = []
    pipe(prompt_embeds=pe, pooled_prompt_embeds=pp, width=128, height=128, output_type='latent',
         callback_on_step_end=lambda p, i, t, kw: (seen.append(i), kw)[1])
    assert seen == [0, 1, 2, 3]
    with pytest.raises(RuntimeError, match='VAE decoder'):
        pipe(prompt_embeds=pe, pooled_prompt_embeds=pp, width=128, height=128, num_inference_steps=2)


@pytest.mark.gpu
def test_qwen_pipeline_runs_vs_oracle():
    from arcflow_amd import FlowMatchEulerDiscreteScheduler
    from arcflow_amd.pipelines import ArcQwenImagePipeline
    from oracle import arcflow_ref as R
    from oracle import dit_ref as D
    cfg = D.QwenCfg(num_layers=2, heads=2, joint_dim=192)
    w = D.make_qwen_weights(cfg, seed=5)
    tcfg = dict(num_layers=2, num_attention_heads=2, attention_head_dim=128, in_channels=64, joint_attention_dim=192)
    pipe = ArcQwenImagePipeline.from_state_dict(tcfg, w, scheduler=FlowMatchEulerDiscreteScheduler(shift=3.2))
    g = torch.Generator().manual_seed(3)
    pe = (torch.randn(1, 20, 192, generator=g) * 0.5).bfloat16()
    mask = torch.zeros(1, 20, dtype=torch.long)
    mask[:, :13] = 1
    lat = torch.randn(1, 64, 64, generator=g)
    out = pipe(prompt_embeds=pe, prompt_embeds_mask=mask, latents=lat, width=128, height=128, num_inference_steps=2,
               timestep_ratio=1.0, output_type='latent', return_dict=False)[0]
    x = lat.clone()
    sig, _ = R.inference_sigmas(2)
    for i in range(2):
        m, lw, lg = D.qwen_forward(w, cfg, x.bfloat16().float(), pe[:, :13].float(), torch.tensor([sig[i]]), 8, 8)
        x = R.momentum_step_packed(x, m, lw, lg, sig[i], sig[i], sig[i + 1])
    err = ((out.cpu() - x).norm() / x.norm()).item()
    assert err < 2.5e-2, err


@pytest.mark.gpu
def test_lora_merge_gpu_vs_host_branch():
    """the device branch of weights._merge_one rounds B * scale and A to bf16 (its GEMM's operand type), the host branch
    forms B A from the fp32 tensors.  bf16 adapters: the two agree to the final rounding; fp32 adapters: the DELTA differs by the
    operand rounding (<= 2^-8 relative in L2), the merged weight by at most one bf16 ulp per entry."""
    from arcflow_amd import weights as W
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(256, 384, generator=g) * 0.02).bfloat16()
    a32, b32 = torch.randn(48, 384, generator=g) * 0.05, torch.randn(256, 48, generator=g) * 0.05
    for a, b, tol in ((a32.bfloat16(), b32.bfloat16(), 1e-6), (a32, b32, 2.0 ** -8)):
        host = W._merge_one(w, a, b, 1.0).float()
        dev = W._merge_one(w.cuda(), a.cuda(), b.cuda(), 1.0).float().cpu()
        exact = w.double() + b.double() @ a.double()
        # both are one bf16 rounding away from their own fp32 sums: compare the sums' deltas through the exact value
        d_host, d_dev = host.double() - w.double(), dev.double() - w.double()
        d_ref = exact - w.double()
        assert ((d_host - d_ref).norm() / d_ref.norm()).item() < 6e-3          # output rounding of W + delta to bf16
        assert ((d_dev - d_ref).norm() / d_ref.norm()).item() < 6e-3 + tol
        if tol < 1e-3:
            assert (host != dev).float().mean().item() < 0.001                 # bf16 adapters: the same sums up to the accumulation order
        else:       # fp32 adapters: entries move by the operand rounding (2^-9 sum |b| |a| ~ 1.5e-4 here) plus at most one bf16 ulp of the result
            assert (host - dev).abs().max().item() < 6e-4 and (host - dev).abs().mean().item() < 5e-5
