"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the block-wise 8-bit AdamW step.

The reference trains with bitsandbytes ``AdamW8bit`` (lakonlab/configs/flux/_ddp_train.py:18-26, lakonlab/runner/optimizer/
builder.py:11-24).  bitsandbytes is not in /root/reference (unvendored, unpinned: requirements.txt:11) -- PARITY UNPINNED: this
follows the published scheme of its block-wise 2-state optimizers (dynamic 8-bit code books, block size 256, one absmax per block,
parameters updated from the un-quantised new moments).  The code books come from arcflow_amd.ops.dynamic_map (host logic, same
restatement); the fp32 recurrences are the pinned AdamW math of oracle/arcflow_ref.py's optimizer tests."""
import torch

BLOCK = 256


def quantize_blockwise(x: torch.Tensor, qmap: torch.Tensor):
    """-> (codes uint8 [n], absmax f32 [ceil(n/256)]): nearest code of x / absmax(block) in the sorted code book."""
    n = x.numel()
    nb = (n + BLOCK - 1) // BLOCK
    pad = torch.zeros(nb * BLOCK, dtype=torch.float32)
    pad[:n] = x.float()
    blocks = pad.view(nb, BLOCK)
    absmax = blocks.abs().amax(dim=1)
    normed = torch.where(absmax[:, None] > 0, blocks / absmax[:, None].clamp(min=1e-45), torch.zeros_like(blocks))
    # nearest code; a tie goes to the lower code (matches the kernel's `x - q[lo-1] <= q[lo] - x`)
    idx = torch.searchsorted(qmap.contiguous(), normed.contiguous(), right=False).clamp(max=255)
    lower = (idx - 1).clamp(min=0)
    take_lower = (idx > 0) & ((normed - qmap[lower]) <= (qmap[idx] - normed))
    codes = torch.where(take_lower, lower, idx)
    return codes.view(-1)[:n].to(torch.uint8), absmax


def dequantize_blockwise(codes: torch.Tensor, absmax: torch.Tensor, qmap: torch.Tensor) -> torch.Tensor:
    n = codes.numel()
    return qmap[codes.long()] * absmax.repeat_interleave(BLOCK)[:n]


def adamw8bit_step(p, g, codes1, codes2, absmax1, absmax2, qmap1, qmap2, lr, step, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0,
                   grad_scale=1.0):
    """One step on fp32 CPU tensors; returns (p, codes1, codes2, absmax1, absmax2).  Same operation order as adamw8bit_kernel."""
    b1, b2 = betas
    g = g.float() * torch.tensor(grad_scale, dtype=torch.float32)
    m = torch.tensor(b1, dtype=torch.float32) * dequantize_blockwise(codes1, absmax1, qmap1) + torch.tensor(1.0 - b1, dtype=torch.float32) * g
    v = torch.tensor(b2, dtype=torch.float32) * dequantize_blockwise(codes2, absmax2, qmap2) + torch.tensor(1.0 - b2, dtype=torch.float32) * g * g
    bc1 = torch.tensor(1.0 - b1 ** step, dtype=torch.float32)
    bc2 = torch.tensor(1.0 - b2 ** step, dtype=torch.float32)
    p = p.float() * torch.tensor(1.0 - lr * weight_decay, dtype=torch.float32)
    p = p - torch.tensor(lr, dtype=torch.float32) * (m / bc1) / ((v / bc2).sqrt() + torch.tensor(eps, dtype=torch.float32))
    c1, a1 = quantize_blockwise(m, qmap1)
    c2, a2 = quantize_blockwise(v, qmap2)
    return p, c1, c2, a1, a2
