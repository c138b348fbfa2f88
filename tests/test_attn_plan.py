"""The KV-split schedule of the one-wave-per-SIMD attention (afx_attn3.hip build_plan): host logic, checked on the CPU through the
library's test hook.  Every (head, sample, 256-query block, key tile) is covered exactly once; whole rounds keep whole blocks; the blocks
of the under-filled last round are cut into one run of key tiles per CU; partial slots of a block are consecutive."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope='module')
def lib():
    from arcflow_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def plan(lib, B, H, S, ncu=256):
    items = (C.c_int * (13 * 8192))()
    comb = (C.c_int * (5 * 4096))()
    nparts, grid = C.c_int(0), C.c_int(0)
    lib.afx_debug_attn_plan.restype = C.c_int
    n = lib.afx_debug_attn_plan(B, H, S, ncu, items, 8192, comb, 4096, C.byref(nparts), C.byref(grid))
    if n <= 0:
        return None
    it = np.frombuffer(items, dtype=np.int32)[:13 * grid.value].reshape(grid.value, 13).copy()
    cb = np.frombuffer(comb, dtype=np.int32)[:5 * n].reshape(n, 5).copy()
    return it, cb, nparts.value, grid.value


@pytest.mark.parametrize('B,H,S', [(1, 24, 4608), (1, 24, 4224), (1, 24, 4173), (2, 24, 4608), (1, 8, 4608), (1, 16, 2048), (1, 3, 1000), (4, 24, 4608)])
def test_plan_covers_every_tile_once(lib, B, H, S):
    p = plan(lib, B, H, S)
    nqb, ntiles = (S + 255) // 256, (S + 63) // 64
    if p is None:        # no XCD has an under-filled last round worth splitting
        per_xcd = [sum(1 for h in range(H) if h % 8 == x) * nqb * B for x in range(8)]
        assert all(n % 32 == 0 or (n % 32) * 8 > 32 * 7 for n in per_xcd) or ntiles < 16
        return
    lmin = max(8, ntiles // 3 + 2)
    it, cb, nparts, grid = p
    assert grid % 8 == 0
    cover = np.zeros((H, B, nqb, ntiles), dtype=np.int32)
    seen_parts = set()
    load = np.zeros(grid, dtype=np.int64)
    for i in range(grid):
        nseg = it[i, 0]
        assert 0 <= nseg <= 2
        for s in range(nseg):
            h, b, qb, t0, n, pidx = it[i, 1 + 6 * s: 7 + 6 * s]
            assert h % 8 == i % 8, 'a head stays on its XCD'
            assert n >= 4 and t0 >= 0 and t0 + n <= ntiles
            cover[h, b, qb, t0:t0 + n] += 1
            load[i] += n
            if n == ntiles:
                assert pidx == -1
            else:
                assert 0 <= pidx < nparts and pidx not in seen_parts
                seen_parts.add(pidx)
    assert (cover == 1).all()
    assert len(seen_parts) == nparts
    # partial slots of a split block are consecutive and listed once
    for h, b, qb, p0, np_ in cb:
        assert 2 <= np_ <= 4
        segs = sorted((t0, n, pidx) for i in range(grid) for s in range(it[i, 0])
                      for (hh, bb, qq, t0, n, pidx) in [it[i, 1 + 6 * s: 7 + 6 * s]] if (hh, bb, qq) == (h, b, qb))
        assert [x[2] for x in segs] == list(range(p0, p0 + np_))
        assert segs[0][0] == 0 and sum(x[1] for x in segs) == ntiles
    assert sum(c[4] for c in cb) == nparts
    # balance: per XCD, the CU that gets slot j of the whole rounds and slot j of the split round carries about the mean load
    for x in range(8):
        lx = load[x::8]
        nblk = sum(1 for h in range(H) if h % 8 == x) * nqb * B
        if nblk % 32 == 0 or (nblk % 32) * 8 > 32 * 7:
            continue
        rem = nblk % 32
        m = min(32, rem * ntiles // lmin)
        if m <= rem:
            continue
        tail = lx[nblk - rem:nblk - rem + m]
        assert tail.max() - tail.min() <= 8 and abs(tail.mean() - rem * ntiles / m) < 1e-6
        assert (lx[nblk - rem + m:] == 0).all()


def test_flux_shape_numbers(lib):
    it, cb, nparts, grid = plan(lib, 1, 24, 4608)
    assert grid == 512                       # two rounds of 256 work-groups
    assert len(cb) == 8 * 22                 # the 22 blocks of each XCD's second round are split
    assert (it[:256, 0] == 1).all() and (it[:256, 5] == 72).all()
    assert it[256:, 5].max() <= 54           # 49.5 key tiles per CU in the second round (+- snapping)
