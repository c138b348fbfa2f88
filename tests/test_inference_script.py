"""The reference's inference script, line for line, on a synthetic snapshot (inference_flux.py:5-30): from_pretrained ->
load_arcflow_adapter(subfolder=...) -> scheduler swap -> .to('cuda') -> pipe(prompt=..., generator=...) -> .images[0].save().
Every stage of the product path runs on the HIP engine (tokenizers are transformers' host code); the result is compared with
the fp32 CPU oracle chain (real transformers text encoders -> dit_ref with the LoRA folded in -> analytic steps -> VAE oracle)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

PROMPT = 'A portrait photo of a kangaroo wearing an orange hoodie and blue sunglasses standing in front of the Sydney Opera House'


def test_inference_flux_script_sequence(tmp_path):
    import snapshot_util as U
    from arcflow_amd import FlowMatchEulerDiscreteScheduler
    from arcflow_amd.pipelines import ArcFluxPipeline
    from oracle import arcflow_ref as R
    from oracle import dit_ref as D
    from oracle import vae_ref as V
    snap = U.write_flux_snapshot(str(tmp_path / 'FLUX.1-dev'))
    ad, lora = U.write_flux_adapter(str(tmp_path / 'ArcFlow'), 'arcflow-flux-2steps', snap)

    # ---- inference_flux.py:5-30 -------------------------------------------------------------------------------------------
    pipe = ArcFluxPipeline.from_pretrained(str(tmp_path / 'FLUX.1-dev'), torch_dtype=torch.bfloat16)
    adapter_name = pipe.load_arcflow_adapter(str(tmp_path / 'ArcFlow'), subfolder='arcflow-flux-2steps', target_module_name='transformer')
    assert adapter_name == 'transformer_arcflow'
    pipe.scheduler = FlowMatchEulerDiscreteScheduler.from_config(pipe.scheduler.config, shift=3.2, shift_terminal=None,
                                                                 use_dynamic_shifting=False)
    pipe = pipe.to('cuda')
    nfe = 2
    seen = []
    out = pipe(prompt=PROMPT, num_images_per_prompt=1, width=128, height=128, num_inference_steps=nfe,
               generator=torch.Generator(device='cuda').manual_seed(42), timestep_ratio=1.0,
               callback_on_step_end=lambda p, i, t, kw: seen.append((i, float(t))) or {}).images[0]
    out.save(str(tmp_path / f'arcflux_{nfe}nfe.png'))
    assert out.size == (128, 128) and os.path.getsize(tmp_path / f'arcflux_{nfe}nfe.png') > 0
    assert [i for i, _ in seen] == [0, 1] and abs(seen[0][1] - 1000.0) < 1e-3

    # ---- the same numbers through the oracles ----------------------------------------------------------------------------------
    from transformers import AutoTokenizer
    tok1 = AutoTokenizer.from_pretrained(str(tmp_path / 'FLUX.1-dev' / 'tokenizer'))
    tok2 = AutoTokenizer.from_pretrained(str(tmp_path / 'FLUX.1-dev' / 'tokenizer_2'))
    with torch.no_grad():
        ids1 = tok1([PROMPT], padding='max_length', max_length=77, truncation=True, return_tensors='pt').input_ids
        ids2 = tok2([PROMPT], padding='max_length', max_length=512, truncation=True, return_tensors='pt').input_ids
        pooled = snap['clip'](ids1).pooler_output.bfloat16().float()
        pe = snap['t5'](ids2)[0].bfloat16().float()
    w = {k: v.float() for k, v in snap['transformer_sd'].items()}
    for k, v in ad.items():
        w[k] = v.float()
    for k in [k for k in lora if '.lora_A.' in k]:
        m = k.rsplit('.lora_A.', 1)[0]
        w[m + '.weight'] = (w[m + '.weight'] + lora[m + '.lora_B.weight'].float() @ lora[k].float()).bfloat16().float()
    hp = wp = 8
    noise = torch.randn((1, 16, 2 * hp, 2 * wp), generator=torch.Generator(device='cuda').manual_seed(42), device='cuda').cpu()
    x = R.pack_latents(noise)
    sig, _ = R.inference_sigmas(nfe)
    for i in range(nfe):
        m, lw, lg = D.flux_forward(w, snap['cfg'], x.bfloat16().float(), pe, pooled, torch.tensor([sig[i]]), torch.tensor([3.5]), hp, wp)
        x = R.momentum_step_packed(x, m, lw, lg, sig[i], sig[i], sig[i + 1])
    ref = V.decode(snap['vae_sd'], (R.unpack_latents(x, hp, wp) / 0.3611 + 0.1159).bfloat16().float(), snap['vae_channels'], groups=16)
    img = pipe(prompt=PROMPT, width=128, height=128, num_inference_steps=nfe, generator=torch.Generator(device='cuda').manual_seed(42),
               timestep_ratio=1.0, output_type='pt').images
    rel = ((img.float().cpu() - ref).norm() / ref.norm()).item()
    assert rel < 5e-2, rel
    # a conditioning handed back by the step callback replaces the prompt embeddings of the following steps (arcflux_pipeline.py:519)
    zero_pe = torch.zeros(1, 512, snap['cfg'].joint_dim)
    img2 = pipe(prompt=PROMPT, width=128, height=128, num_inference_steps=nfe, generator=torch.Generator(device='cuda').manual_seed(42),
                timestep_ratio=1.0, output_type='pt', callback_on_step_end=lambda p, i, t, kw: {'prompt_embeds': zero_pe},
                callback_on_step_end_tensor_inputs=['latents', 'prompt_embeds']).images
    assert (img2.float() - img.float()).abs().max().item() > 1e-3


QWEN_PROMPT = ('A semi-realistic fantasy illustration featuring a split composition of two young men in profile, facing away from each other. '
               'On the left, a pale man with sharp features and black hair wears a dark coat. On the right, a tan man wears a blue tunic.')


def test_inference_qwen_script_sequence(tmp_path):
    """/root/reference/inference_qwen.py:5-30 line for line on a synthetic Qwen/Qwen-Image snapshot: from_pretrained ->
    load_arcflow_adapter(subfolder='arcflow-qwen-2steps') -> scheduler swap -> .to('cuda') -> pipe(prompt=...).images[0].save()."""
    import snapshot_util as U
    from arcflow_amd import FlowMatchEulerDiscreteScheduler
    from arcflow_amd.pipelines import ArcQwenImagePipeline
    from oracle import arcflow_ref as R
    from oracle import dit_ref as D
    from oracle import vae_qwen_ref as V
    snap = U.write_qwen_snapshot(str(tmp_path / 'Qwen-Image'))
    ad, lora = U.write_qwen_adapter(str(tmp_path / 'ArcFlow'), 'arcflow-qwen-2steps', snap)

    # ---- inference_qwen.py:5-30 -------------------------------------------------------------------------------------------
    pipe = ArcQwenImagePipeline.from_pretrained(str(tmp_path / 'Qwen-Image'), torch_dtype=torch.bfloat16)
    adapter_name = pipe.load_arcflow_adapter(str(tmp_path / 'ArcFlow'), subfolder='arcflow-qwen-2steps', target_module_name='transformer')
    assert adapter_name == 'transformer_arcflow'
    pipe.scheduler = FlowMatchEulerDiscreteScheduler.from_config(pipe.scheduler.config, shift=3.2, shift_terminal=None,
                                                                 use_dynamic_shifting=False)
    pipe = pipe.to('cuda')
    nfe = 2
    out = pipe(prompt=QWEN_PROMPT, num_images_per_prompt=1, width=128, height=128, num_inference_steps=nfe,
               generator=torch.Generator(device='cuda').manual_seed(42), timestep_ratio=1.0).images[0]
    out.save(str(tmp_path / f'arcqwen_{nfe}nfe.png'))
    assert out.size == (128, 128) and os.path.getsize(tmp_path / f'arcqwen_{nfe}nfe.png') > 0

    # ---- the same numbers through the oracles: transformers language model -> dit_ref with the LoRA folded in -> analytic steps -> VAE oracle
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(str(tmp_path / 'Qwen-Image' / 'tokenizer'))
    drop = pipe.prompt_template_encode_start_idx
    enc = tok([pipe.prompt_template_encode.format(QWEN_PROMPT)], max_length=pipe.tokenizer_max_length + drop, padding=True, truncation=True,
              return_tensors='pt')
    with torch.no_grad():
        hidden = snap['lm'](input_ids=enc.input_ids, attention_mask=enc.attention_mask, output_hidden_states=True).hidden_states[-1]
    pe = hidden[0][enc.attention_mask[0].bool()][drop:][None].bfloat16().float()
    assert pe.shape[1] > 8                                        # the prompt survives the 34-token template drop
    w = {k: v.float() for k, v in snap['transformer_sd'].items() if not k.startswith('proj_out.')}
    for k, v in ad.items():
        w[k] = v.float()
    for k in [k for k in lora if '.lora_A.' in k]:
        m = k.rsplit('.lora_A.', 1)[0]
        w[m + '.weight'] = (w[m + '.weight'] + lora[m + '.lora_B.weight'].float() @ lora[k].float()).bfloat16().float()
    hp = wp = 8
    noise = torch.randn((1, 16, 2 * hp, 2 * wp), generator=torch.Generator(device='cuda').manual_seed(42), device='cuda').cpu()
    x = R.pack_latents(noise)
    sig, _ = R.inference_sigmas(nfe)
    for i in range(nfe):
        m, lw, lg = D.qwen_forward(w, snap['cfg'], x.bfloat16().float(), pe, torch.tensor([sig[i]]), hp, wp)
        x = R.momentum_step_packed(x, m, lw, lg, sig[i], sig[i], sig[i + 1])
    z = R.unpack_latents(x, hp, wp) * torch.tensor(snap['latents_std']).view(1, 16, 1, 1) + torch.tensor(snap['latents_mean']).view(1, 16, 1, 1)
    ref = V.decode(snap['vae_sd'], z.bfloat16().float())
    img = pipe(prompt=QWEN_PROMPT, width=128, height=128, num_inference_steps=nfe, generator=torch.Generator(device='cuda').manual_seed(42),
               timestep_ratio=1.0, output_type='pt').images
    rel = ((img.float().cpu() - ref).norm() / ref.norm()).item()
    assert rel < 5e-2, rel
    # the base snapshot alone (no adapter) refuses to sample, as the reference's teacher-headed transformer would fail in policy construction
    base = ArcQwenImagePipeline.from_pretrained(str(tmp_path / 'Qwen-Image'), torch_dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match='load_arcflow_adapter'):
        base(prompt=QWEN_PROMPT, width=128, height=128, num_inference_steps=2)
