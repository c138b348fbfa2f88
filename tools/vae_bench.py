#!/usr/bin/env python3
"""1024^2 decode time of the two VAE decoders (random weights of the released shapes)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd.vae import AutoencoderKLDecoder, AutoencoderKLQwenImageDecoder  # noqa: E402
from oracle import vae_qwen_ref, vae_ref  # noqa: E402   (weight generators only)


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


tok = torch.randn(1, 4096, 64, device='cuda')
which = sys.argv[1] if len(sys.argv) > 1 else 'both'
if which in ('both', 'flux'):
  flux = AutoencoderKLDecoder(vae_ref.make_decoder_weights((128, 256, 512, 512), seed=0), (128, 256, 512, 512))
  print(f'FLUX AutoencoderKL decoder        1024^2: {timeit(lambda: flux.decode_packed(tok, 64, 64)):.1f} ms')
if which in ('both', 'qwen'):
  qwen = AutoencoderKLQwenImageDecoder(vae_qwen_ref.make_decoder_weights(dim=96, seed=0), [0.0] * 16, [1.0] * 16)
  print(f'Qwen AutoencoderKLQwenImage decoder 1024^2: {timeit(lambda: qwen.decode_packed(tok, 64, 64)):.1f} ms')
