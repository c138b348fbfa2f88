#!/usr/bin/env python3
"""Pre-encode prompts into the distillation cache (the reference's "Data Preparation" step, configs/flux/README.md):

    python tools/cache_prompts.py --family flux --snapshot /path/to/FLUX.1-dev --prompts prompts.txt --out-dir data/preproc_flux

prompts.txt: one prompt per line.  One process per GPU under torchrun shards the lines rank-strided.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--family', choices=['flux', 'qwen'], required=True)
    ap.add_argument('--snapshot', required=True, help='local model snapshot with text_encoder*/ and tokenizer*/')
    ap.add_argument('--prompts', required=True)
    ap.add_argument('--out-dir', required=True)
    ap.add_argument('--max-sequence-length', type=int, default=512)
    ap.add_argument('--latent-size', type=int, nargs=3, default=[16, 128, 128])
    ap.add_argument('--batch', type=int, default=8)
    args = ap.parse_args()
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    from arcflow_amd.train.prompts import PromptEncoder, write_cache
    enc = PromptEncoder.from_snapshot(args.family, args.snapshot, max_sequence_length=args.max_sequence_length)
    with open(args.prompts, encoding='utf-8') as f:
        lines = [l.rstrip('\n') for l in f if l.strip()]
    mine = lines[rank::world]
    out = args.out_dir if world == 1 else os.path.join(args.out_dir, f'rank{rank}')
    names = write_cache(enc, mine, out, tuple(args.latent_size), args.batch)
    print(f'[rank {rank}] wrote {len(names)} items to {out}')


if __name__ == '__main__':
    main()
