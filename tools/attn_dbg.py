"""bring-up aid of the one-wave-per-SIMD attention kernel: one launch at a small shape under AFX_ATTN3_DBG=n (the work-group leaves
after stage n), prints whether the launch completed and the output error (only meaningful for n = 0)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from arcflow_amd import ops  # noqa: E402

S = int(os.environ.get('S', '128'))
H = int(os.environ.get('H', '1'))
g = torch.Generator(device='cuda').manual_seed(1)
q, k, v = (torch.randn(1, S, H, 128, generator=g, device='cuda').bfloat16() for _ in range(3))
ops.set_attn_impl(0)
print('launching, dbg =', os.environ.get('AFX_ATTN3_DBG', '0'), 'S', S, 'H', H, flush=True)
out = ops.attention(q, k, v)
torch.cuda.synchronize()
ref = torch.nn.functional.scaled_dot_product_attention(q.float().transpose(1, 2), k.float().transpose(1, 2),
                                                       v.float().transpose(1, 2)).transpose(1, 2).reshape(1, S, H * 128)
err = ((out.float() - ref).norm() / ref.norm()).item()
print(f'completed; rel-L2 {err:.3e} finite {bool(torch.isfinite(out.float()).all())}', flush=True)
