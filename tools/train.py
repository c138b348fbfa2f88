#!/usr/bin/env python3
"""Data-free ArcFlow distillation from one of the reference's config files (same command line as its train.py:47-94):

    python tools/train.py configs/flux/arcflux_2nfe_k16.py --diff_seed
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py <config> --launcher pytorch --diff_seed

One process per GPU; the gradient exchange is the RCCL all-reduce of the trainable set (arcflow_amd/train/reducer.py).
Weights: ``--transformer-dir`` = a local diffusers ``transformer/`` directory of the teacher (FLUX.1-dev / Qwen-Image);
``--synthetic`` = random-init weights of the configured architecture (no network in this image).  Prompt embeddings:
``--data-dir`` = the reference's preprocessed cache (arcflow_amd/train/data.py) or synthetic ones.
Checkpoints are written / resumed in the reference's ``iter_N.pth`` layout (arcflow_amd/train/checkpoint.py).
"""
import argparse
import ast
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_args():
    ap = argparse.ArgumentParser(description='ArcFlow distillation on MI355X')
    ap.add_argument('config')
    ap.add_argument('--work-dir')
    ap.add_argument('--resume-from')
    ap.add_argument('--seed', type=int, default=2021)
    ap.add_argument('--diff_seed', action='store_true', help='different seeds for different ranks')
    ap.add_argument('--cfg-options', nargs='+', default=[], help='key=value overrides of the config')
    ap.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    ap.add_argument('--iters', type=int, help='stop after this many iterations (default: total_iters of the config)')
    ap.add_argument('--transformer-dir', help='local diffusers transformer directory of the teacher')
    ap.add_argument('--synthetic', action='store_true', help='random-init weights and prompt embeddings')
    ap.add_argument('--data-dir', help='prompt-embedding cache directory')
    ap.add_argument('--negative-prompt-embeds', help='torch.load-able embeddings of the negative prompt (true-CFG teacher: Qwen config; '
                                                     'the reference dataset option negative_prompt_embeds_path)')
    ap.add_argument('--prompts', help='text file, one prompt per line: encode on the fly with the HIP text encoders (needs --snapshot)')
    ap.add_argument('--snapshot', help='local FLUX.1-dev / Qwen-Image snapshot (text_encoder*/, tokenizer*/) for --prompts')
    ap.add_argument('--latent-tokens', type=int, nargs=2, default=[64, 64], help='synthetic data: packed latent grid (64 64 = 1024^2)')
    ap.add_argument('--export', help='write the EMA adapter (diffusers layout) here when done')
    return ap.parse_args()


def _options(pairs):
    out = {}
    for p in pairs:
        k, v = p.split('=', 1)
        try:
            out[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[k] = v
    return out


def main():
    # the host driver only supports dmabuf IPC: must be in the environment BEFORE the first torch.cuda call initialises the HIP / HSA runtime
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    args = parse_args()
    from arcflow_amd.train import ArcFlowDistiller, checkpoint, config, data
    cfg = config.apply_options(config.load_config(args.config), _options(args.cfg_options))
    family, eng, dc, run = config.distill_setup(cfg)
    rank, world, local = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))
    pg = None
    if args.launcher == 'pytorch' or world > 1:
        from arcflow_amd.train import init_distributed
        dist, dev = init_distributed(local)           # picks this rank's device, then RCCL: rank r on GPU r (train.py:182-185 init_dist('pytorch', backend='nccl'))
        pg = dist.group.WORLD
    else:
        torch.cuda.set_device(local)
        dev = f'cuda:{torch.cuda.current_device()}'
    seed = args.seed + (rank if args.diff_seed else 0)              # train.py:222-225
    rng = torch.Generator(device=dev).manual_seed(seed)

    # ---- weights --------------------------------------------------------------------------------------
    if args.synthetic:
        from arcflow_amd.weights import random_packed
        D = eng['heads'] * eng['head_dim']
        packed = random_packed(family, eng['num_double'], eng.get('num_single', 0), dev, heads=eng['heads'], in_channels=eng['in_channels'],
                               joint_dim=eng['joint_dim'], pooled_dim=eng.get('pooled_dim', 768), guidance=eng.get('guidance_embeds', True),
                               K=eng['num_gaussians'], L=eng['logweights_channels'], seed=0)
        g = torch.Generator(device=dev).manual_seed(1)
        packed['teacher_head.weight'] = (torch.randn(eng['in_channels'], D, generator=g, device=dev) * 0.02).bfloat16()
        packed['teacher_head.bias'] = torch.zeros(eng['in_channels'], device=dev, dtype=torch.bfloat16)
        packed['norm_out.weight'] = packed['mod.weight'][-2 * D:].clone()
        packed['norm_out.bias'] = packed['mod.bias'][-2 * D:].clone()
        dist_ = ArcFlowDistiller(family, eng, None, dc, device=dev, process_group=pg, packed=packed)
    else:
        if not args.transformer_dir:
            raise SystemExit(f"--transformer-dir is required (config names {run['pretrained']!r}; this image has no network) -- or pass --synthetic")
        from arcflow_amd.pipelines.arcflux_pipeline import load_transformer_dir
        from arcflow_amd.weights import init_arcflow_heads_from_teacher
        _, sd = load_transformer_dir(args.transformer_dir)
        sd = init_arcflow_heads_from_teacher(sd, K=eng['num_gaussians'], L=eng['logweights_channels'], generator=torch.Generator().manual_seed(args.seed))
        dist_ = ArcFlowDistiller(family, eng, sd, dc, device=dev, process_group=pg)

    # ---- resume (resume_from of the config = checkpoints/<name>/latest.pth; with --work-dir the latest checkpoint under it) ----
    ckpt_dir = os.path.join(args.work_dir, 'checkpoints') if args.work_dir else run['ckpt_dir']
    resume = args.resume_from or (os.path.join(ckpt_dir, 'latest.pth') if args.work_dir else run['resume_from'])
    if resume and os.path.exists(resume):
        meta = checkpoint.load_checkpoint(dist_, resume)
        if rank == 0:
            print(f'[train] resumed from {resume} at iteration {meta.get("iter")}')
    elif run['load_from'] and os.path.exists(run['load_from']):
        checkpoint.load_checkpoint(dist_, run['load_from'])
        dist_.iteration = 0

    # ---- data ---------------------------------------------------------------------------------------------
    B = run['samples_per_gpu']
    loader = None
    if args.data_dir:
        neg_path = args.negative_prompt_embeds or run['data_train'].get('negative_prompt_embeds_path')
        if dc.teacher_guidance_scale > 1.0 and not neg_path:
            raise SystemExit(f'teacher_guidance_scale = {dc.teacher_guidance_scale} (true CFG) needs the negative prompt embeddings: '
                             'pass --negative-prompt-embeds <file> (or data.train.negative_prompt_embeds_path in the config)')
        ds = data.PromptEmbedCache(args.data_dir, pad_seq_len=512 if family == 'flux' else None, bucketize=True,
                                   negative_prompt_embeds_path=neg_path)
        sampler = data.DistributedSampler(ds, world, rank, shuffle=True, samples_per_gpu=B, seed=args.seed)
        sampler.set_iter(dist_.iteration)

        def batches():
            epoch = dist_.iteration // max(1, sampler.num_samples // B)
            while True:
                sampler.set_epoch(epoch)
                idx = list(iter(sampler))
                for i in range(0, len(idx) - B + 1, B):
                    yield data.collate([ds[j] for j in idx[i:i + B]], device=dev)
                epoch += 1
        loader = batches()
    elif args.prompts:          # text encoder inside the training process (the uncommented text_encoder of _ddp_train.py)
        if not args.snapshot:
            raise SystemExit('--prompts needs --snapshot')
        from arcflow_amd.train.prompts import PromptEncoder
        enc = PromptEncoder.from_snapshot(family, args.snapshot)
        with open(args.prompts, encoding='utf-8') as f:
            lines = [l.rstrip('\n') for l in f if l.strip()]
        neg = [' '] * B if dc.teacher_guidance_scale > 1.0 else None

        def prompt_batches():
            i = rank * B + dist_.iteration * B * world
            while True:
                yield enc.cond([lines[(i + j) % len(lines)] for j in range(B)], args.latent_tokens[0], args.latent_tokens[1], neg)
                i += B * world
        loader = prompt_batches()
    else:
        T = 512 if family == 'flux' else 128
        synth = dict(prompt_embeds=(torch.randn(B, T, eng['joint_dim'], device=dev, generator=rng) * 0.1).bfloat16(), hp=args.latent_tokens[0], wp=args.latent_tokens[1])
        if family == 'flux':
            synth['pooled'] = (torch.randn(B, eng['pooled_dim'], device=dev, generator=rng) * 0.1).bfloat16()
        if dc.teacher_guidance_scale > 1.0:
            synth['negative_prompt_embeds'] = (torch.randn(B, T, eng['joint_dim'], device=dev, generator=rng) * 0.1).bfloat16()

    total = args.iters if args.iters is not None else run['total_iters']
    t_last = time.perf_counter()
    while dist_.iteration < total:
        cond = next(loader) if loader is not None else synth
        info = dist_.train_step(cond, B, rng=rng)
        # a background save that failed stops the run now -- on EVERY rank: only the writer's rank sees the error, the others would otherwise walk into the
        # next gradient all-reduce and sit there until the RCCL timeout.  One scalar MAX-reduce per iteration (an iteration is seconds).
        if dist_.reducer.all_reduce_max(1.0 if checkpoint.pending_save_failed() else 0.0, dev) > 0:
            checkpoint.check_pending_save()
            raise RuntimeError(f'rank {rank}: the checkpoint writer of another rank failed (see its log): stopping')
        if rank == 0:
            now = time.perf_counter()
            print(json.dumps(dict(iter=dist_.iteration, time=round(now - t_last, 3), **{k: (round(v, 6) if isinstance(v, float) else v) for k, v in info.items()})), flush=True)
            t_last = now
        if dist_.iteration % run['save_interval'] == 0 or dist_.iteration == total:
            dist_.reducer.check_consistent(dist_.params)      # every rank took the same optimizer steps: raises before a diverged state is saved
            if rank == 0:       # host copy here, pickling + the file write in a thread (the Qwen-Image adapter set is ~6 GB: 14 s of disk time per save)
                path = checkpoint.save_checkpoint_async(dist_, ckpt_dir, fp16=run['ckpt_fp16'], fp16_ema=run['ckpt_fp16_ema'])
                print(f'[train] saving {path}', flush=True)
    checkpoint.wait_pending_save()
    if args.export and rank == 0:
        print('[train] exported', checkpoint.export_adapter(dist_, args.export, ema=True, policy_kwargs=run['policy_kwargs']))
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
