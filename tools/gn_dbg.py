import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import _lib
from arcflow_amd.vae import _Grid, _p, _s
lib = _lib.load()
H, W, ci, co, groups = 9, 13, 64, 128, 32
g = torch.Generator().manual_seed(1)
x = torch.randn(ci, H, W, generator=g).bfloat16()
wt = (torch.randn(co, ci, 3, 3, generator=g) * 0.05).bfloat16()
b = torch.randn(co, generator=g).bfloat16()
gx, gy = _Grid(H, W, ci, 'cuda'), _Grid(H, W, co, 'cuda')
gx.t.view(H + 2, W + 2, ci)[1:-1, 1:-1] = x.permute(1, 2, 0).cuda()
wp = wt.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous().cuda()
slots = torch.zeros(64, groups, 2, dtype=torch.float64, device='cuda')
_lib.check(lib.afx_conv3x3_bf16_stats(_p(gx.t), _p(wp), _p(b.cuda()), _p(gy.t), H, W, ci, co, None, _p(slots), groups, _s()))
got = slots.sum(0).cpu()
y = gy.t.view(H + 2, W + 2, co).double().cpu()
yg = y.view(-1, groups, co // groups)
ref = torch.stack([yg.sum((0, 2)), (yg * yg).sum((0, 2))], 1)
torch.set_printoptions(linewidth=200, precision=3)
print('got', got[:8].t()); print('ref', ref[:8].t())
print('nonzero slots', (slots.abs().sum((1, 2)) > 0).nonzero().flatten().tolist())
print('per-channel sums', y.view(-1, co).sum(0)[:16])
