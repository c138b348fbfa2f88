"""The balanced schedule of the one-wave-per-SIMD attention's last round (afx_attn3.hip build_plan): host logic, checked on the CPU through
the library's test hook.  Every (head, sample, 256-query block, key tile) is covered exactly once; the blocks of the under-filled last round are
cut once (long part [0, L) + short end [L, ntiles)); a long part names exactly the partial slots its block's short pieces write, and those pieces come
EARLIER in the XCD's dispatch order; a greedy list schedule of the items ends close to (blocks x tiles) / CUs."""
import ctypes as C

import numpy as np
import pytest

W = 1 + 8 * 8


@pytest.fixture(scope='module')
def lib():
    from arcflow_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def plan(lib, B, H, S, ncu=256):
    items = (C.c_int * (W * 16384))()
    nparts, grid = C.c_int(0), C.c_int(0)
    lib.afx_debug_attn_plan.restype = C.c_int
    n = lib.afx_debug_attn_plan(B, H, S, ncu, items, 16384, C.byref(nparts), C.byref(grid))
    if n <= 0:
        return None
    it = np.frombuffer(items, dtype=np.int32)[:W * grid.value].reshape(grid.value, W).copy()
    return it, nparts.value, grid.value


def segs(it, i):
    return [tuple(it[i, 1 + 8 * s: 9 + 8 * s]) for s in range(it[i, 0])]


def worth_splitting(nblk, ntiles):
    R, rem = divmod(nblk, 32)
    if not (rem >= 4 and R >= 1 and rem * 4 <= 32 * 3 and ntiles >= 16):
        return False
    L = (rem * (ntiles + 7) - 2 * (32 - rem)) // 32           # the cut point: whole + long = short run + whole, segment overheads folded in
    return L >= 8 and ntiles - L >= 8 and 8 - (rem // 4 + (1 if rem % 4 else 0)) >= 1


def simulate(it, x, grid, cost):
    """The dispatcher tools/dispatch_probe.hip shows: the j-th work-group of an XCD goes to engine j % 4 (8 CUs) and waits IN ORDER -- blocking
    the ones behind it -- until that engine has a free CU.  Returns the makespan of XCD x's list."""
    free = np.zeros((4, 8))
    now = 0.0
    j = 0
    for i in range(x, grid, 8):
        sg = segs(it, i)
        if not sg:
            continue
        e = j % 4
        j += 1
        c = int(np.argmin(free[e]))
        now = max(now, free[e, c])
        free[e, c] = now + cost(sg)
    return free.max()


@pytest.mark.parametrize('B,H,S', [(1, 24, 4608), (1, 24, 4224), (1, 24, 4173), (2, 24, 4608), (1, 8, 4608), (1, 16, 2048), (1, 3, 1000), (4, 24, 4608),
                                   (1, 40, 4608), (3, 24, 1024)])
def test_plan_covers_every_tile_once(lib, B, H, S):
    p = plan(lib, B, H, S)
    nqb, ntiles = (S + 255) // 256, (S + 63) // 64
    per_xcd = [sum(1 for h in range(H) if h % 8 == x) * nqb * B for x in range(8)]
    if p is None:
        assert not any(worth_splitting(n, ntiles) for n in per_xcd)
        return
    it, nparts, grid = p
    assert grid % 8 == 0
    cover = np.zeros((H, B, nqb, ntiles), dtype=np.int32)
    writer = {}                      # partial slot -> (item index, block)
    for i in range(grid):
        assert 0 <= it[i, 0] <= 8
        for (h, b, qb, t0, n, out, nin, in0) in segs(it, i):
            assert h % 8 == i % 8, 'a head stays on its XCD'
            assert n >= 4 and t0 >= 0 and t0 + n <= ntiles
            cover[h, b, qb, t0:t0 + n] += 1
            if out >= 0:
                assert 0 <= out < nparts and out not in writer and nin == 0
                writer[out] = (i, (h, b, qb))
    assert (cover == 1).all()
    assert len(writer) == nparts
    used = set()
    for i in range(grid):
        for (h, b, qb, t0, n, out, nin, in0) in segs(it, i):
            if nin == 0:
                continue
            assert out == -1 and t0 == 0 and 1 <= nin <= 4
            for s in range(in0, in0 + nin):
                wi, blk = writer[s]
                assert blk == (h, b, qb), 'a long part starts from ITS block\'s partials'
                assert wi % 8 == i % 8 and wi < i, 'written earlier in the same XCD\'s dispatch order'
                assert s not in used
                used.add(s)
            # long part + its short pieces = the whole key range
            assert n + sum(sg[4] for j in range(grid) for sg in segs(it, j) if sg[5] in range(in0, in0 + nin)) == ntiles
    assert used == set(writer)
    # makespan under the in-order, engine-rotating dispatcher; cost = key tiles + a per-segment overhead
    def cost(sg):
        return sum(s_[4] + (7 if s_[5] >= 0 else 3 + (2 if s_[6] else 0)) for s_ in sg)
    for x in range(8):
        if not worth_splitting(per_xcd[x], ntiles):
            continue
        span = simulate(it, x, grid, cost)
        ideal = per_xcd[x] * (ntiles + 3) / 32
        plain = -(-per_xcd[x] // 32) * (ntiles + 3)
        assert span <= 1.18 * ideal + 6, (span, ideal, plain)
        assert span < plain - 0.3 * (plain - ideal), (span, ideal, plain)


def test_flux_shape_numbers(lib):
    it, nparts, grid = plan(lib, 1, 24, 4608)
    assert grid == 8 * 64
    x0 = [segs(it, i) for i in range(0, grid, 8)]
    # engine k = slots k, k + 4, ...: [rem_k whole] [8 - rem_k runs of short ends] [8 - rem_k whole] [rem_k long parts], rem_k = 5, 5, 6, 6
    for k, rk in enumerate((5, 5, 6, 6)):
        sub = x0[k::4]
        assert len(sub) == 16
        assert all(len(s) == 1 and s[0][4] == 72 and s[0][5] == -1 and s[0][6] == 0 for s in sub[:rk] + sub[8:16 - rk])
        assert all(1 <= len(s) <= 8 and all(g[5] >= 0 for g in s) for s in sub[rk:8])
        L = sub[16 - rk][0][4]
        assert L == x0[60][0][4] and 48 <= L <= 58 and all(len(s) == 1 and s[0][4] == L and s[0][6] >= 1 and s[0][3] == 0 for s in sub[16 - rk:])
    assert nparts >= 8 * 22
