#!/usr/bin/env python3
"""Prompt-encoder latency at the released sizes (random weights generated on the GPU): T5 v1.1 XXL encoder at 512 tokens,
CLIP ViT-L/14 text model at 77, Qwen2.5-VL-7B language model at 34 + 128 tokens."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd.text_encoders import CLIPTextEncoder, Qwen25TextEncoder, T5Encoder  # noqa: E402

dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)


def rn(*shape, std=0.02):
    return (torch.randn(*shape, device=dev, generator=g) * std).bfloat16()


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def t5_sd(L=24, D=4096, F=10240, H=64, dk=64, V=32128):
    sd = {'shared.weight': rn(V, D, std=1.0), 'encoder.final_layer_norm.weight': torch.ones(D, device=dev),
          'encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight': rn(32, H, std=0.5).float()}
    for i in range(L):
        a, f = f'encoder.block.{i}.layer.0.', f'encoder.block.{i}.layer.1.'
        for n in 'qkv':
            sd[a + f'SelfAttention.{n}.weight'] = rn(H * dk, D)
        sd[a + 'SelfAttention.o.weight'] = rn(D, H * dk)
        sd[a + 'layer_norm.weight'] = torch.ones(D, device=dev)
        sd[f + 'DenseReluDense.wi_0.weight'], sd[f + 'DenseReluDense.wi_1.weight'] = rn(F, D), rn(F, D)
        sd[f + 'DenseReluDense.wo.weight'] = rn(D, F)
        sd[f + 'layer_norm.weight'] = torch.ones(D, device=dev)
    return sd


def clip_sd(L=12, D=768, F=3072, V=49408):
    sd = {'embeddings.token_embedding.weight': rn(V, D), 'embeddings.position_embedding.weight': rn(77, D),
          'final_layer_norm.weight': torch.ones(D, device=dev), 'final_layer_norm.bias': torch.zeros(D, device=dev)}
    for i in range(L):
        p = f'encoder.layers.{i}.'
        for n in 'qkv':
            sd[p + f'self_attn.{n}_proj.weight'], sd[p + f'self_attn.{n}_proj.bias'] = rn(D, D), rn(D)
        sd[p + 'self_attn.out_proj.weight'], sd[p + 'self_attn.out_proj.bias'] = rn(D, D), rn(D)
        sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'] = rn(F, D), rn(F)
        sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'] = rn(D, F), rn(D)
        for n in ('layer_norm1', 'layer_norm2'):
            sd[p + n + '.weight'], sd[p + n + '.bias'] = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    return sd


def qwen_sd(L=28, D=3584, F=18944, H=28, Hkv=4, V=152064):
    d = D // H
    sd = {'model.embed_tokens.weight': rn(V, D), 'model.norm.weight': torch.ones(D, device=dev)}
    for i in range(L):
        p = f'model.layers.{i}.'
        sd[p + 'self_attn.q_proj.weight'], sd[p + 'self_attn.q_proj.bias'] = rn(H * d, D), rn(H * d)
        for n in 'kv':
            sd[p + f'self_attn.{n}_proj.weight'], sd[p + f'self_attn.{n}_proj.bias'] = rn(Hkv * d, D), rn(Hkv * d)
        sd[p + 'self_attn.o_proj.weight'] = rn(D, H * d)
        sd[p + 'mlp.gate_proj.weight'], sd[p + 'mlp.up_proj.weight'], sd[p + 'mlp.down_proj.weight'] = rn(F, D), rn(F, D), rn(D, F)
        sd[p + 'input_layernorm.weight'] = sd[p + 'post_attention_layernorm.weight'] = torch.ones(D, device=dev)
    return sd


def main():
    t5 = T5Encoder(t5_sd())
    ids = torch.randint(0, 32000, (1, 512))
    print(f'T5 v1.1 XXL encoder, 512 tokens:        {timeit(lambda: t5(ids)):7.2f} ms')
    del t5
    clip = CLIPTextEncoder(clip_sd(), eos_token_id=2)
    ids = torch.randint(0, 49000, (1, 77))
    print(f'CLIP ViT-L/14 text model, 77 tokens:    {timeit(lambda: clip(ids)):7.2f} ms')
    del clip
    qw = Qwen25TextEncoder(qwen_sd())
    ids = torch.randint(0, 150000, (1, 162))
    print(f'Qwen2.5-VL-7B language model, 162 tok:  {timeit(lambda: qw(ids)):7.2f} ms')


if __name__ == '__main__':
    main()
