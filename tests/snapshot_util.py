"""Writes a SYNTHETIC diffusers snapshot + ArcFlow adapter in the on-disk layout the reference's inference scripts consume
(inference_flux.py:5-12: ``ArcFluxPipeline.from_pretrained('black-forest-labs/FLUX.1-dev')`` then
``load_arcflow_adapter('ymyy307/ArcFlow', subfolder='arcflow-flux-2steps')``): reduced width, random weights, real file
formats (config.json, sharded safetensors + index, scheduler_config.json, tokenizer.json).  Test infrastructure only."""
import json
import os

import torch


def _save_sharded(sd, folder, nshards=2):
    from safetensors.torch import save_file
    os.makedirs(folder, exist_ok=True)
    keys = sorted(sd)
    per = (len(keys) + nshards - 1) // nshards
    weight_map = {}
    for s in range(nshards):
        part = {k: sd[k].contiguous() for k in keys[s * per:(s + 1) * per]}
        name = f'diffusion_pytorch_model-{s + 1:05d}-of-{nshards:05d}.safetensors'
        save_file(part, os.path.join(folder, name))
        weight_map.update({k: name for k in part})
    with open(os.path.join(folder, 'diffusion_pytorch_model.safetensors.index.json'), 'w') as f:
        json.dump({'metadata': {'total_size': sum(v.numel() * v.element_size() for v in sd.values())}, 'weight_map': weight_map}, f)


def _tiny_tokenizer(folder, max_len, clip_style):
    """A word-level fast tokenizer with the special tokens the two FLUX tokenizers use (CLIP: bos/eos, pad = eos, eos has
    the LARGEST id -- CLIPTextModel pools at argmax(ids); T5: pad 0, eos 1)."""
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    words = ['a', 'portrait', 'photo', 'of', 'kangaroo', 'wearing', 'an', 'orange', 'hoodie', 'and', 'blue', 'sunglasses',
             'standing', 'in', 'front', 'the', 'sydney', 'opera', 'house', 'holding', 'sign', 'on', 'chest', 'that', 'says',
             'welcome', 'friends', 'cat', 'dog', 'red']
    if clip_style:
        vocab = {'<unk>': 0, '<|startoftext|>': 1}
        vocab.update({w: i + 2 for i, w in enumerate(words)})
        vocab['<|endoftext|>'] = len(vocab)
        tok = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
        tok.pre_tokenizer = pre_tokenizers.Whitespace()
        tok.post_processor = processors.TemplateProcessing(single='<|startoftext|> $A <|endoftext|>',
                                                           special_tokens=[('<|startoftext|>', 1), ('<|endoftext|>', vocab['<|endoftext|>'])])
        fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token='<|startoftext|>', eos_token='<|endoftext|>',
                                       pad_token='<|endoftext|>', unk_token='<unk>', model_max_length=max_len)
    else:
        vocab = {'<pad>': 0, '</s>': 1, '<unk>': 2}
        vocab.update({w: i + 3 for i, w in enumerate(words)})
        tok = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
        tok.pre_tokenizer = pre_tokenizers.Whitespace()
        tok.post_processor = processors.TemplateProcessing(single='$A </s>', special_tokens=[('</s>', 1)])
        fast = PreTrainedTokenizerFast(tokenizer_object=tok, eos_token='</s>', pad_token='<pad>', unk_token='<unk>',
                                       model_max_length=max_len)
    fast.save_pretrained(folder)
    return len(vocab)


def write_flux_snapshot(root, num_layers=1, num_single_layers=1, heads=2, joint_dim=128, pooled_dim=64, with_text=True,
                        with_vae=True, seed=0):
    """-> dict(transformer_sd, vae_sd, cfg, vae_channels, t5, clip) of what was written (fp32/bf16 CPU tensors, HF modules)."""
    from oracle import dit_ref as D
    from oracle import vae_ref as V
    cfg = D.FluxCfg(num_layers=num_layers, num_single_layers=num_single_layers, heads=heads, joint_dim=joint_dim, pooled_dim=pooled_dim)
    w = D.make_flux_weights(cfg, seed=seed, teacher_head=True)
    teacher = {k: v for k, v in w.items() if not k.startswith('proj_out_')}
    tdir = os.path.join(root, 'transformer')
    _save_sharded(teacher, tdir, nshards=3)
    tcfg = {'_class_name': 'FluxTransformer2DModel', '_diffusers_version': '0.35.1', 'attention_head_dim': 128,
            'axes_dims_rope': [16, 56, 56], 'guidance_embeds': True, 'in_channels': 64, 'joint_attention_dim': joint_dim,
            'num_attention_heads': heads, 'num_layers': num_layers, 'num_single_layers': num_single_layers, 'patch_size': 1,
            'pooled_projection_dim': pooled_dim}
    json.dump(tcfg, open(os.path.join(tdir, 'config.json'), 'w'))
    os.makedirs(os.path.join(root, 'scheduler'), exist_ok=True)
    json.dump({'_class_name': 'FlowMatchEulerDiscreteScheduler', '_diffusers_version': '0.35.1', 'base_image_seq_len': 256,
               'base_shift': 0.5, 'max_image_seq_len': 4096, 'max_shift': 1.15, 'num_train_timesteps': 1000, 'shift': 3.0,
               'use_dynamic_shifting': True}, open(os.path.join(root, 'scheduler', 'scheduler_config.json'), 'w'))
    out = dict(cfg=cfg, transformer_cfg=tcfg, transformer_sd=w)
    if with_vae:
        from safetensors.torch import save_file
        chans = (64, 128, 128, 128)
        vw = V.make_decoder_weights(chans, seed=seed + 1)
        os.makedirs(os.path.join(root, 'vae'), exist_ok=True)
        save_file({k: v.contiguous() for k, v in vw.items()}, os.path.join(root, 'vae', 'diffusion_pytorch_model.safetensors'))
        json.dump({'_class_name': 'AutoencoderKL', 'block_out_channels': list(chans), 'norm_num_groups': 16, 'layers_per_block': 2,
                   'latent_channels': 16, 'scaling_factor': 0.3611, 'shift_factor': 0.1159},
                  open(os.path.join(root, 'vae', 'config.json'), 'w'))
        out.update(vae_sd=vw, vae_channels=chans)
    if with_text:
        from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel
        torch.manual_seed(seed + 2)
        v_clip = _tiny_tokenizer(os.path.join(root, 'tokenizer'), 77, clip_style=True)
        v_t5 = _tiny_tokenizer(os.path.join(root, 'tokenizer_2'), 512, clip_style=False)
        clip = CLIPTextModel(CLIPTextConfig(vocab_size=v_clip, hidden_size=pooled_dim, intermediate_size=128, num_hidden_layers=2,
                                            num_attention_heads=1, max_position_embeddings=77, hidden_act='quick_gelu',
                                            bos_token_id=1, eos_token_id=v_clip - 1, pad_token_id=v_clip - 1)).eval()
        t5 = T5EncoderModel(T5Config(vocab_size=v_t5, d_model=joint_dim, d_kv=64, d_ff=256, num_layers=2, num_heads=2,
                                     feed_forward_proj='gated-gelu', dropout_rate=0.0)).eval()
        for m in (clip, t5):
            for p in m.parameters():
                p.data = p.data.bfloat16().float()
        clip.save_pretrained(os.path.join(root, 'text_encoder'), safe_serialization=True)
        t5.save_pretrained(os.path.join(root, 'text_encoder_2'), safe_serialization=True)
        out.update(clip=clip, t5=t5)
    return out


def write_flux_adapter(root, subfolder, snap, rank=16, seed=5):
    """ArcFlow adapter directory as export_arcflow_to_diffusers.py:100-127 writes it: config.json (_class_name + constructor
    arguments) + diffusion_pytorch_model.safetensors holding the three heads, norm_out and the LoRA pairs (``lora_A.weight``
    names, target prefix stripped), metadata ``policy_config``.  -> (adapter state dict, LoRA dict)"""
    from safetensors.torch import save_file
    w, cfg = snap['transformer_sd'], snap['cfg']
    g = torch.Generator().manual_seed(seed)
    ad = {k: v.clone() for k, v in w.items() if k.startswith(('proj_out_', 'norm_out.'))}
    ad['norm_out.linear.weight'] = (ad['norm_out.linear.weight'].float() + 0.01 * torch.randn(ad['norm_out.linear.weight'].shape, generator=g)).bfloat16()
    lora = {}
    targets = [f'transformer_blocks.{i}.{ff}.{n}' for i in range(cfg.num_layers) for ff in ('ff', 'ff_context') for n in ('net.0.proj', 'net.2')]
    targets += [f'single_transformer_blocks.{i}.{n}' for i in range(cfg.num_single_layers) for n in ('proj_mlp', 'proj_out')]
    targets += ['time_text_embed.timestep_embedder.linear_1', 'time_text_embed.timestep_embedder.linear_2']
    for t in targets:
        o, i = w[t + '.weight'].shape
        lora[t + '.lora_A.weight'] = (torch.randn(rank, i, generator=g) / rank).bfloat16()
        lora[t + '.lora_B.weight'] = (torch.randn(o, rank, generator=g) * 0.02).bfloat16()
    d = os.path.join(root, subfolder)
    os.makedirs(d, exist_ok=True)
    full = dict(ad)
    full.update(lora)
    save_file({k: v.contiguous() for k, v in full.items()}, os.path.join(d, 'diffusion_pytorch_model.safetensors'),
              metadata={'policy_config': json.dumps({'type': 'ArcFlow'})})
    acfg = dict(snap['transformer_cfg'])
    acfg.update({'_class_name': 'ArcFluxTransformer2DModel', 'num_gaussians': 16, 'logweights_channels': 4})
    json.dump(acfg, open(os.path.join(d, 'config.json'), 'w'))
    return ad, lora


# ---------------------------------------------------------------------------------------------------------------- Qwen-Image
def _qwen_tokenizer(folder, max_len=2048):
    """A word-level fast tokenizer that knows the chat-template markers of the Qwen-Image prompt template (right padding)."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    words = ('system user assistant Describe the image by detailing color shape size texture quantity text spatial relationships of '
             'objects and background a semi realistic fantasy illustration featuring split composition two young men in profile facing '
             'away from each other on left pale man with sharp features black hair wears dark coat right tan blue tunic red teal '
             'painterly brushstrokes cat dog').split()
    vocab = {'<|endoftext|>': 0, '<unk>': 1, '<|im_start|>': 2, '<|im_end|>': 3, ',': 4, ':': 5, '.': 6}
    for w in words:
        vocab.setdefault(w, len(vocab))
    tok = Tokenizer(models.WordLevel(vocab, unk_token='<unk>'))
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split('<|im_start|>', 'isolated'), pre_tokenizers.Split('<|im_end|>', 'isolated'),
                                                 pre_tokenizers.Whitespace()])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token='<|endoftext|>', unk_token='<unk>', model_max_length=max_len,
                                   padding_side='right')
    fast.save_pretrained(folder)
    return len(vocab)


def write_qwen_snapshot(root, num_layers=2, heads=2, joint_dim=256, vae_dim=32, with_text=True, with_vae=True, seed=0):
    """Synthetic Qwen/Qwen-Image snapshot in the layout inference_qwen.py:5-8 loads: transformer/ (QwenImageTransformer2DModel:
    sharded safetensors + index + config.json), vae/ (AutoencoderKLQwenImage), text_encoder/ (Qwen2_5_VLForConditionalGeneration),
    tokenizer/, scheduler/.  -> dict of what was written."""
    from oracle import dit_ref as D
    from oracle import vae_qwen_ref as V
    cfg = D.QwenCfg(num_layers=num_layers, heads=heads, joint_dim=joint_dim)
    w = D.make_qwen_weights(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 11)
    teacher = {k: v for k, v in w.items() if not k.startswith('proj_out_')}
    teacher['proj_out.weight'] = (torch.randn(cfg.in_channels, cfg.dim, generator=g) * 0.02).bfloat16()       # the base model's velocity head
    teacher['proj_out.bias'] = (torch.randn(cfg.in_channels, generator=g) * 0.02).bfloat16()
    tdir = os.path.join(root, 'transformer')
    _save_sharded(teacher, tdir, nshards=3)
    tcfg = {'_class_name': 'QwenImageTransformer2DModel', '_diffusers_version': '0.35.1', 'attention_head_dim': 128,
            'axes_dims_rope': [16, 56, 56], 'guidance_embeds': False, 'in_channels': 64, 'joint_attention_dim': joint_dim,
            'num_attention_heads': heads, 'num_layers': num_layers, 'out_channels': 16, 'patch_size': 2}
    json.dump(tcfg, open(os.path.join(tdir, 'config.json'), 'w'))
    os.makedirs(os.path.join(root, 'scheduler'), exist_ok=True)
    json.dump({'_class_name': 'FlowMatchEulerDiscreteScheduler', '_diffusers_version': '0.35.1', 'base_image_seq_len': 256,
               'base_shift': 0.5, 'max_image_seq_len': 8192, 'max_shift': 0.9, 'num_train_timesteps': 1000, 'shift': 1.0,
               'shift_terminal': 0.02, 'use_dynamic_shifting': True}, open(os.path.join(root, 'scheduler', 'scheduler_config.json'), 'w'))
    out = dict(cfg=cfg, transformer_cfg=tcfg, transformer_sd=dict(w, **{k: teacher[k] for k in ('proj_out.weight', 'proj_out.bias')}))
    if with_vae:
        from safetensors.torch import save_file
        vw = V.make_decoder_weights(dim=vae_dim, seed=seed + 1)
        mean = (torch.randn(16, generator=g) * 0.3).tolist()
        std = (1.0 + 0.5 * torch.rand(16, generator=g)).tolist()
        os.makedirs(os.path.join(root, 'vae'), exist_ok=True)
        save_file({k: v.contiguous() for k, v in vw.items()}, os.path.join(root, 'vae', 'diffusion_pytorch_model.safetensors'))
        json.dump({'_class_name': 'AutoencoderKLQwenImage', 'base_dim': vae_dim, 'dim_mult': [1, 2, 4, 4], 'num_res_blocks': 2, 'z_dim': 16,
                   'latents_mean': mean, 'latents_std': std, 'temperal_downsample': [False, True, True]},
                  open(os.path.join(root, 'vae', 'config.json'), 'w'))
        out.update(vae_sd=vw, latents_mean=mean, latents_std=std)
    if with_text:
        from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
        torch.manual_seed(seed + 2)
        nv = _qwen_tokenizer(os.path.join(root, 'tokenizer'))
        assert joint_dim % 128 == 0           # the language model's head dim must be 128 (mrope_section [16, 24, 24])
        text = dict(vocab_size=nv, hidden_size=joint_dim, intermediate_size=256, num_hidden_layers=2, num_attention_heads=joint_dim // 128, num_key_value_heads=1,
                    max_position_embeddings=2048, rope_theta=1e6, rms_norm_eps=1e-6, tie_word_embeddings=False,
                    rope_scaling=dict(type='mrope', mrope_section=[16, 24, 24]))
        vis = dict(depth=1, hidden_size=64, intermediate_size=128, num_heads=2, out_hidden_size=joint_dim)
        try:
            mcfg = Qwen2_5_VLConfig(text_config=text, vision_config=vis)
        except TypeError:
            mcfg = Qwen2_5_VLConfig(vision_config=vis, **text)
        lm = Qwen2_5_VLForConditionalGeneration(mcfg).eval()
        for p in lm.parameters():
            p.data = p.data.bfloat16().float()
        lm.save_pretrained(os.path.join(root, 'text_encoder'), safe_serialization=True)
        out.update(lm=lm)
    return out


def write_qwen_adapter(root, subfolder, snap, rank=16, seed=5):
    """ArcFlow-Qwen adapter directory (export_arcflow_to_diffusers.py:100-127; LoRA targets of configs/qwen/arcqwen_2nfe_k16.py:47-58:
    img_mlp of every block, txt_mlp of every block but the last, the timestep-embedder pair).  -> (adapter state dict, LoRA dict)"""
    from safetensors.torch import save_file
    w, cfg = snap['transformer_sd'], snap['cfg']
    g = torch.Generator().manual_seed(seed)
    ad = {k: v.clone() for k, v in w.items() if k.startswith(('proj_out_', 'norm_out.'))}
    ad['norm_out.linear.weight'] = (ad['norm_out.linear.weight'].float() + 0.01 * torch.randn(ad['norm_out.linear.weight'].shape, generator=g)).bfloat16()
    targets = [f'transformer_blocks.{i}.img_mlp.{n}' for i in range(cfg.num_layers) for n in ('net.0.proj', 'net.2')]
    targets += [f'transformer_blocks.{i}.txt_mlp.{n}' for i in range(cfg.num_layers - 1) for n in ('net.0.proj', 'net.2')]
    targets += ['time_text_embed.timestep_embedder.linear_1', 'time_text_embed.timestep_embedder.linear_2']
    lora = {}
    for t in targets:
        o, i = w[t + '.weight'].shape
        lora[t + '.lora_A.weight'] = (torch.randn(rank, i, generator=g) / rank).bfloat16()
        lora[t + '.lora_B.weight'] = (torch.randn(o, rank, generator=g) * 0.02).bfloat16()
    d = os.path.join(root, subfolder)
    os.makedirs(d, exist_ok=True)
    full = dict(ad)
    full.update(lora)
    save_file({k: v.contiguous() for k, v in full.items()}, os.path.join(d, 'diffusion_pytorch_model.safetensors'),
              metadata={'policy_config': json.dumps({'type': 'ArcFlow'})})
    acfg = dict(snap['transformer_cfg'])
    acfg.update({'_class_name': 'ArcQwenImageTransformer2DModel', 'num_gaussians': 16, 'logweights_channels': 4})
    json.dump(acfg, open(os.path.join(d, 'config.json'), 'w'))
    return ad, lora
