#!/usr/bin/env python3
"""End-to-end latency of one 1024^2 image on one MI355X with random weights of the released sizes:
token ids -> T5-XXL + CLIP-L -> 2-NFE FLUX denoiser + ArcFlow steps -> AutoencoderKL decode.  (Tokenisation is host-side
string work and needs the snapshot's vocabulary files, so the run starts from token ids.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import ops  # noqa: E402
from arcflow_amd.text_encoders import CLIPTextEncoder, T5Encoder  # noqa: E402
from arcflow_amd.vae import AutoencoderKLDecoder  # noqa: E402
from bench import N_IMG, build_flux_engine  # noqa: E402
from oracle import vae_ref  # noqa: E402   (weight generator only)
from tools.text_bench import clip_sd, t5_sd  # noqa: E402


def main():
    dev = 'cuda'
    t5, clip = T5Encoder(t5_sd()), CLIPTextEncoder(clip_sd(), eos_token_id=2)
    eng, (_, _, _, _, guidance, hp, wp) = build_flux_engine('flux', dev)
    vae = AutoencoderKLDecoder(vae_ref.make_decoder_weights((128, 256, 512, 512), seed=0), (128, 256, 512, 512))
    sig = [1.0, 0.7619047619, 0.0]
    ids5, idsc = torch.randint(0, 32000, (1, 512)), torch.randint(0, 49000, (1, 77))
    lat = torch.randn(1, N_IMG, 64, device=dev)
    tv = [torch.full((1,), s, device=dev) for s in sig[:2]]

    def run(stamps=None):
        def mark(name):
            if stamps is not None:
                torch.cuda.synchronize()
                stamps.append((name, time.perf_counter()))
        mark('start')
        pe = t5(ids5)
        pooled = clip(idsc)[1]
        mark('text encoders')
        x = lat
        for i in range(2):
            out = eng(x.bfloat16(), tv[i], pe, pooled, guidance, hp, wp)
            x = ops.arcflow_step(x, out.means, out.logweights, out.loggammas, sig[i], sig[i], sig[i + 1])
        mark('denoiser 2 NFE')
        img = vae.decode_packed(x, hp, wp)
        mark('vae decode')
        return img

    for _ in range(2):
        run()
    tot = {}
    n = 5
    for _ in range(n):
        st = []
        run(st)
        for (a, ta), (b, tb) in zip(st[:-1], st[1:]):
            tot[b] = tot.get(b, 0.0) + (tb - ta)
    total = sum(tot.values()) / n
    for k, v in tot.items():
        print(f'{k:18s} {v / n * 1e3:7.1f} ms')
    print(f'{"prompt -> image":18s} {total * 1e3:7.1f} ms  = {1 / total:.2f} images/s (single stream, 1 GPU)')


if __name__ == '__main__':
    main()
