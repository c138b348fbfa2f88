"""Block-wise 8-bit AdamW (the reference's optimizer class is bitsandbytes AdamW8bit: _ddp_train.py:18-26).  bitsandbytes is an
unvendored dependency: the code books and the block scheme are a restatement (parity unpinned); what IS checked: the code books'
structure, that the 8-bit trajectory tracks the pinned fp32 AdamW math, and that the HIP kernel equals the CPU restatement."""
import pytest
import torch

from arcflow_amd.ops import dynamic_map
from oracle import adamw8bit_ref as R


def test_dynamic_code_books():
    s, u = dynamic_map(True), dynamic_map(False)
    for q in (s, u):
        assert q.shape == (256,) and q.dtype == torch.float32 and bool((q[1:] > q[:-1]).all()) and q[-1] == 1.0
    assert torch.equal(s[:127].flip(0), -s[128:255]) and s[127] == 0.0           # mirrored: 127 negative, 0, 127 positive, then 1.0
    assert float(s[0]) == pytest.approx(-0.9929687, rel=1e-6)                    # midpoint of the last of 64 cells of [0.1, 1]
    assert u[0] == 0.0 and float(u[1]) == pytest.approx(3.25e-7, rel=1e-5)       # first cell of [0.1, 1] * 1e-6 cut in two
    # exponent i contributes 2^i (signed) / 2^(i+1) (unsigned) codes in (10^(i-7), 10^(i-6)]
    for i in range(7):
        lo, hi = 10.0 ** (i - 7), 10.0 ** (i - 6)
        assert int(((s > lo) & (s < hi)).sum()) == 2 ** i and int(((u > lo) & (u < hi)).sum()) == 2 ** (i + 1)


def test_blockwise_quantisation_round_trip_error():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1000, generator=g) * torch.logspace(-4, 0, 1000)
    for q, y in ((dynamic_map(True), x), (dynamic_map(False), x * x)):
        c, a = R.quantize_blockwise(y, q)
        assert c.dtype == torch.uint8 and a.shape == (4,)
        d = R.dequantize_blockwise(c, a, q)
        rel = (d - y).abs() / y.abs().clamp(min=1e-30)
        am = a.repeat_interleave(256)[:1000]
        # decade i of the book has 2^i (2^(i+1)) cells of [0.1, 1] * 10^(i-6): half a cell at the low end of the top decade is
        # 0.45 / 64 / 0.1 = 7 % (signed), of the next decade 14 %
        assert float(rel[y.abs() > 0.1 * am].max()) < 0.0705 and float(rel[y.abs() > 0.01 * am].max()) < 0.141
        assert torch.equal(R.quantize_blockwise(d, q)[0], c)                                      # idempotent


def _problem(n, seed):
    g = torch.Generator().manual_seed(seed)
    target = torch.randn(n, generator=g)
    scale = torch.logspace(-2, 0, n)
    return target, scale


def test_8bit_trajectory_tracks_fp32_adamw():
    n, steps, lr = 1500, 60, 1e-2
    target, scale = _problem(n, 1)
    q1, q2 = dynamic_map(True), dynamic_map(False)
    p8 = torch.zeros(n)
    c1 = c2 = torch.zeros(n, dtype=torch.uint8)
    a1 = a2 = torch.zeros(6)
    p32, m, v = torch.zeros(n), torch.zeros(n), torch.zeros(n)
    for t in range(1, steps + 1):
        g8 = scale * (p8 - target)
        p8, c1, c2, a1, a2 = R.adamw8bit_step(p8, g8, c1, c2, a1, a2, q1, q2, lr, t)
        g32 = scale * (p32 - target)
        m = 0.9 * m + 0.1 * g32
        v = 0.95 * v + 0.05 * g32 * g32
        p32 = p32 - lr * (m / (1 - 0.9 ** t)) / ((v / (1 - 0.95 ** t)).sqrt() + 1e-8)
        if t == 1:
            assert torch.allclose(p8, p32, atol=1e-7)          # the parameter is updated from the UNquantised new moments
    # per-element moment errors of a few % (see the round-trip test) turn into a few % of path difference over 60 steps ...
    assert float((p8 - p32).norm() / p32.norm()) < 0.12
    # ... while the optimisation progress is the same
    e8, e32 = float((p8 - target).norm()), float((p32 - target).norm())
    assert e8 < 0.75 * float(target.norm()) and abs(e8 - e32) < 0.05 * e32


@pytest.mark.gpu
@pytest.mark.parametrize('n', [256, 1000, 70000])
def test_hip_adamw8bit_step_matches_restatement(n):
    from arcflow_amd import ops
    g = torch.Generator().manual_seed(n)
    p = torch.randn(n, generator=g)
    st = ops.AdamW8bitState(n, 'cuda')
    pd = p.cuda()
    q1, q2 = dynamic_map(True), dynamic_map(False)
    c1 = c2 = torch.zeros(n, dtype=torch.uint8)
    a1 = a2 = torch.zeros((n + 255) // 256)
    for t in range(1, 5):
        grad = torch.randn(n, generator=g) * torch.logspace(-3, 0, n)
        ops.adamw8bit_step(pd, grad.cuda(), st, 1e-3, t, betas=(0.9, 0.95), weight_decay=0.01, grad_scale=0.5)
        p, c1, c2, a1, a2 = R.adamw8bit_step(p, grad, c1, c2, a1, a2, q1, q2, 1e-3, t, weight_decay=0.01, grad_scale=0.5)
        torch.cuda.synchronize()
        assert torch.allclose(pd.cpu(), p, rtol=2e-5, atol=1e-7), t
        assert torch.allclose(st.absmax1.cpu(), a1, rtol=1e-6) and torch.allclose(st.absmax2.cpu(), a2, rtol=1e-6)
        # codes: identical except where fp32 rounding puts a value on the other side of a cell boundary
        assert float((st.state1.cpu() != c1).float().mean()) < 2e-3 and float((st.state2.cpu() != c2).float().mean()) < 2e-3
        assert int((st.state1.cpu().int() - c1.int()).abs().max()) <= 1
        c1, c2 = st.state1.cpu(), st.state2.cpu()               # continue from the kernel's codes so single flips do not accumulate
    m, v = st.moments()
    assert torch.allclose(m.cpu(), R.dequantize_blockwise(c1, a1, q1)) and torch.allclose(v.cpu(), R.dequantize_blockwise(c2, a2, q2))
