#!/usr/bin/env python3
"""Distillation-step throughput on one MI355X (BASELINE.json configs[3] shape per GPU: FLUX-12B architecture,
4 samples, 1024^2 latents, 512 text tokens, random-init weights).  Trainable set of this round: heads + norm_out.

    python tools/train_bench.py [--batch 4] [--iters 3] [--model flux]
    torchrun --nproc-per-node N tools/train_bench.py ...     (data parallel, RCCL all-reduce of the trainables)
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--model', default='flux')
    ap.add_argument('--lora-rank', type=int, default=256)
    ap.add_argument('--lora-dropout', type=float, default=0.05, help='peft lora_dropout of the reference configs')
    ap.add_argument('--teacher-fp8', action='store_true', help='BASELINE configs[4]: teacher forwards on the fp8 MFMA')
    args = ap.parse_args()
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    dev = f'cuda:{local}'
    from arcflow_amd.train import ArcFlowDistiller, DistillConfig
    from arcflow_amd.weights import random_packed
    assert args.model in ('flux', 'qwen')
    D = 3072
    flux = args.model == 'flux'
    nd, ns, joint, T = (19, 38, 4096, 512) if flux else (60, 0, 3584, 128)
    packed = random_packed(args.model, nd, ns, dev, joint_dim=joint, seed=0)
    g = torch.Generator(device=dev).manual_seed(1)
    packed['teacher_head.weight'] = (torch.randn(64, D, generator=g, device=dev) * 0.02).bfloat16()
    packed['teacher_head.bias'] = torch.zeros(64, device=dev, dtype=torch.bfloat16)
    packed['norm_out.weight'] = packed['mod.weight'][-2 * D:].clone()
    packed['norm_out.bias'] = packed['mod.bias'][-2 * D:].clone()
    # configs/qwen/arcqwen_2nfe_k16.py: true-CFG teacher (scale 4.0, negative prompt), decay 1000, batch 2 per GPU
    dc = DistillConfig(lora_rank=args.lora_rank, lora_dropout=args.lora_dropout, teacher_fp8=args.teacher_fp8) if flux else \
        DistillConfig(lora_rank=args.lora_rank, lora_dropout=args.lora_dropout, teacher_guidance_scale=4.0, num_decay_iters=1000, teacher_fp8=args.teacher_fp8)
    eng = dict(num_double=nd, num_single=ns) if flux else dict(num_double=nd, joint_dim=joint)
    dist_ = ArcFlowDistiller(args.model, eng, None, dc, device=dev, packed=packed)
    B = args.batch
    cond = dict(prompt_embeds=(torch.randn(B, T, joint, device=dev, generator=g) * 0.1).bfloat16(), hp=64, wp=64)
    if flux:
        cond['pooled'] = (torch.randn(B, 768, device=dev, generator=g) * 0.1).bfloat16()
    else:
        cond['negative_prompt_embeds'] = (torch.randn(B, T, joint, device=dev, generator=g) * 0.1).bfloat16()
    rng = torch.Generator(device=dev).manual_seed(100 + rank)
    for _ in range(args.warmup):
        info = dist_.train_step(cond, B, rng=rng)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        info = dist_.train_step(cond, B, rng=rng)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    # student fwd 2 + teacher fwd 8 (+ per-block recompute 2 + backward ~4 with the LoRA trunk): SURVEY 3.3 counts 16
    fwd_equiv = (16 if flux else 24) if args.lora_rank > 0 else (10 if flux else 18)     # Qwen's CFG teacher doubles the 8
    if rank == 0:
        print(json.dumps({'metric': 'distillation samples/s (' + ('LoRA r=%d + ' % args.lora_rank if args.lora_rank else '') + 'heads + norm_out trainable)', 'value': world * B / dt,
                          's_per_iter': dt, 'batch_per_gpu': B, 'n_gpus': world, 'last': info,
                          'forward_equivalents_per_sample': fwd_equiv, 'trainable_params': int(dist_.params.numel()),
                          'max_mem_gb': torch.cuda.max_memory_allocated() / 2 ** 30,
                          'denoiser_tflops': world * B * fwd_equiv * (74.41 if flux else 70.6) / dt}))


if __name__ == '__main__':
    main()
