import sys, torch
sys.path.insert(0, '/root/repo')
from arcflow_amd import ops
for (M, N, K) in [(256, 256, 64), (256, 256, 128), (256, 256, 256), (512, 512, 512), (4608, 3072, 3072)]:
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    out = ops.linear(a, w, None).float()
    ref = a.float() @ w.float().T
    d = (out - ref).abs()
    bad = d > 0.02 * ref.abs().max()
    print(M, N, K, 'rel', ((out - ref).norm() / ref.norm()).item(), 'bad frac', bad.float().mean().item())
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print('  bad rows', rows[:10].tolist(), '...', rows.numel(), ' bad cols', cols[:10].tolist(), '...', cols.numel())
        # is it a k-half problem? compare with partial products
        for name, ks in (('k<32', slice(0, 32)), ('k 32..64', slice(32, 64)), ('k<64', slice(0, 64))):
            r2 = a.float()[:, ks] @ w.float()[:, ks].T
            print('   vs', name, ((out - r2).norm() / r2.norm()).item())
