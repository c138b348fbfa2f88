"""Random-init state dicts with the key names / shapes of the released checkpoints, drawn on the device, for the networks
AROUND the denoiser (prompt encoders, VAE decoders): what `bench.py`'s end-to-end object and the latency tools feed the product
classes when no checkpoint can be downloaded.  (`weights.random_packed` does the same for the denoisers.)  Key layouts:
transformers' T5EncoderModel / CLIPTextModel / Qwen2_5_VL language model, diffusers' AutoencoderKL / AutoencoderKLQwenImage
(the classes the reference pipelines hold: lakonlab/pipelines/arcflux_pipeline.py:104-133, arcqwen_pipeline.py:80-104)."""
from __future__ import annotations

from typing import Dict, Sequence

import torch

Tensor = torch.Tensor


def _gen(device, seed):
    return torch.Generator(device=device).manual_seed(seed)


def t5_state_dict(device='cuda', seed=0, L=24, D=4096, F=10240, H=64, dk=64, V=32128) -> Dict[str, Tensor]:
    g = _gen(device, seed)

    def rn(*shape, std=0.02):
        return (torch.randn(*shape, device=device, generator=g) * std).bfloat16()
    sd = {'shared.weight': rn(V, D, std=1.0), 'encoder.final_layer_norm.weight': torch.ones(D, device=device),
          'encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight': rn(32, H, std=0.5).float()}
    for i in range(L):
        a, f = f'encoder.block.{i}.layer.0.', f'encoder.block.{i}.layer.1.'
        for n in 'qkv':
            sd[a + f'SelfAttention.{n}.weight'] = rn(H * dk, D)
        sd[a + 'SelfAttention.o.weight'] = rn(D, H * dk)
        sd[a + 'layer_norm.weight'] = torch.ones(D, device=device)
        sd[f + 'DenseReluDense.wi_0.weight'], sd[f + 'DenseReluDense.wi_1.weight'] = rn(F, D), rn(F, D)
        sd[f + 'DenseReluDense.wo.weight'] = rn(D, F)
        sd[f + 'layer_norm.weight'] = torch.ones(D, device=device)
    return sd


def clip_state_dict(device='cuda', seed=1, L=12, D=768, F=3072, V=49408) -> Dict[str, Tensor]:
    g = _gen(device, seed)

    def rn(*shape, std=0.02):
        return (torch.randn(*shape, device=device, generator=g) * std).bfloat16()
    sd = {'embeddings.token_embedding.weight': rn(V, D), 'embeddings.position_embedding.weight': rn(77, D),
          'final_layer_norm.weight': torch.ones(D, device=device), 'final_layer_norm.bias': torch.zeros(D, device=device)}
    for i in range(L):
        p = f'encoder.layers.{i}.'
        for n in 'qkv':
            sd[p + f'self_attn.{n}_proj.weight'], sd[p + f'self_attn.{n}_proj.bias'] = rn(D, D), rn(D)
        sd[p + 'self_attn.out_proj.weight'], sd[p + 'self_attn.out_proj.bias'] = rn(D, D), rn(D)
        sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'] = rn(F, D), rn(F)
        sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'] = rn(D, F), rn(D)
        for n in ('layer_norm1', 'layer_norm2'):
            sd[p + n + '.weight'], sd[p + n + '.bias'] = torch.ones(D, device=device), torch.zeros(D, device=device)
    return sd


def qwen25_state_dict(device='cuda', seed=2, L=28, D=3584, F=18944, H=28, Hkv=4, V=152064) -> Dict[str, Tensor]:
    g = _gen(device, seed)

    def rn(*shape, std=0.02):
        return (torch.randn(*shape, device=device, generator=g) * std).bfloat16()
    d = D // H
    sd = {'model.embed_tokens.weight': rn(V, D), 'model.norm.weight': torch.ones(D, device=device)}
    for i in range(L):
        p = f'model.layers.{i}.'
        sd[p + 'self_attn.q_proj.weight'], sd[p + 'self_attn.q_proj.bias'] = rn(H * d, D), rn(H * d)
        for n in 'kv':
            sd[p + f'self_attn.{n}_proj.weight'], sd[p + f'self_attn.{n}_proj.bias'] = rn(Hkv * d, D), rn(Hkv * d)
        sd[p + 'self_attn.o_proj.weight'] = rn(D, H * d)
        sd[p + 'mlp.gate_proj.weight'], sd[p + 'mlp.up_proj.weight'], sd[p + 'mlp.down_proj.weight'] = rn(F, D), rn(F, D), rn(D, F)
        sd[p + 'input_layernorm.weight'] = sd[p + 'post_attention_layernorm.weight'] = torch.ones(D, device=device)
    return sd


def vae_kl_decoder_state_dict(device='cuda', seed=3, block_out_channels: Sequence[int] = (128, 256, 512, 512), latent_channels=16,
                              layers_per_block=2) -> Dict[str, Tensor]:
    """diffusers AutoencoderKL decoder (FLUX.1-dev: 16 latent channels, (128, 256, 512, 512))."""
    g = _gen(device, seed)
    w: Dict[str, Tensor] = {}

    def rn(*shape):
        return torch.randn(*shape, device=device, generator=g)

    def conv(name, co, ci, k=3):
        w[name + '.weight'] = (rn(co, ci, k, k) * (1.2 / (ci * k * k) ** 0.5)).bfloat16()
        w[name + '.bias'] = (rn(co) * 0.05).bfloat16()

    def norm(name, c):
        w[name + '.weight'] = (1 + 0.1 * rn(c)).bfloat16()
        w[name + '.bias'] = (0.1 * rn(c)).bfloat16()

    def res(p, ci, co):
        norm(p + 'norm1', ci); conv(p + 'conv1', co, ci); norm(p + 'norm2', co); conv(p + 'conv2', co, co)
        if ci != co:
            conv(p + 'conv_shortcut', co, ci, 1)
    rev = list(reversed(block_out_channels))
    c0 = rev[0]
    conv('decoder.conv_in', c0, latent_channels)
    res('decoder.mid_block.resnets.0.', c0, c0)
    p = 'decoder.mid_block.attentions.0.'
    norm(p + 'group_norm', c0)
    for nm in ('to_q', 'to_k', 'to_v', 'to_out.0'):
        w[p + nm + '.weight'] = (rn(c0, c0) * (1.0 / c0 ** 0.5)).bfloat16()
        w[p + nm + '.bias'] = (rn(c0) * 0.05).bfloat16()
    res('decoder.mid_block.resnets.1.', c0, c0)
    prev = c0
    for i, co in enumerate(rev):
        for j in range(layers_per_block + 1):
            res(f'decoder.up_blocks.{i}.resnets.{j}.', prev if j == 0 else co, co)
        prev = co
        if i < len(rev) - 1:
            conv(f'decoder.up_blocks.{i}.upsamplers.0.conv', co, co)
    norm('decoder.conv_norm_out', prev)
    conv('decoder.conv_out', 3, prev)
    return w


def vae_qwen_decoder_state_dict(device='cuda', seed=4, dim=96, z_dim=16, dim_mult: Sequence[int] = (1, 2, 4, 4), num_res_blocks=2,
                                temporal_upsample: Sequence[bool] = (False, True, True), std=0.03) -> Dict[str, Tensor]:
    """diffusers AutoencoderKLQwenImage decoder + post_quant_conv (causal conv3d weights, RMS-norm gammas)."""
    g = _gen(device, seed)
    w: Dict[str, Tensor] = {}

    def rn(*shape):
        return torch.randn(*shape, device=device, generator=g)

    def conv3(name, co, ci, k):
        w[name + '.weight'] = rn(co, ci, k, k, k) * (std if k == 3 else std * 3)
        w[name + '.bias'] = rn(co) * 0.02

    def conv2(name, co, ci, k):
        w[name + '.weight'] = rn(co, ci, k, k) * (std if k == 3 else std * 2)
        w[name + '.bias'] = rn(co) * 0.02

    def norm(name, c, images):
        w[name + '.gamma'] = (1.0 + 0.1 * rn(c)).reshape((c, 1, 1) if images else (c, 1, 1, 1))

    def res(p, ci, co):
        norm(p + 'norm1', ci, False); conv3(p + 'conv1', co, ci, 3)
        norm(p + 'norm2', co, False); conv3(p + 'conv2', co, co, 3)
        if ci != co:
            conv3(p + 'conv_shortcut', co, ci, 1)
    dims = [dim * u for u in [dim_mult[-1]] + list(dim_mult[::-1])]
    conv3('post_quant_conv', z_dim, z_dim, 1)
    conv3('decoder.conv_in', dims[0], z_dim, 3)
    res('decoder.mid_block.resnets.0.', dims[0], dims[0])
    a = 'decoder.mid_block.attentions.0.'
    norm(a + 'norm', dims[0], True); conv2(a + 'to_qkv', dims[0] * 3, dims[0], 1); conv2(a + 'proj', dims[0], dims[0], 1)
    res('decoder.mid_block.resnets.1.', dims[0], dims[0])
    for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
        if i > 0:
            ci = ci // 2
        cur = ci
        for j in range(num_res_blocks + 1):
            res(f'decoder.up_blocks.{i}.resnets.{j}.', cur, co)
            cur = co
        if i != len(dim_mult) - 1:
            conv2(f'decoder.up_blocks.{i}.upsamplers.0.resample.1', co // 2, co, 3)
            if temporal_upsample[i]:
                w[f'decoder.up_blocks.{i}.upsamplers.0.time_conv.bias'] = rn(co * 2) * 0.02
                w[f'decoder.up_blocks.{i}.upsamplers.0.time_conv.weight'] = rn(co * 2, co, 3, 1, 1) * std
    norm('decoder.norm_out', dims[-1], False)
    conv3('decoder.conv_out', 3, dims[-1], 3)
    return w
