#!/usr/bin/env python3
"""Which half of ops.attention is racy: the V^T re-layout or the flash kernel?  Compares both outputs across launches."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcflow_amd import _lib  # noqa: E402

lib = _lib.load()
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
g = torch.Generator(device='cuda').manual_seed(0)
for S, H in ((4608, 24), (4173, 24), (1101, 24)):
    q, k, v = (torch.randn(S, H * 128, generator=g, device='cuda').bfloat16() for _ in range(3))
    o = torch.empty_like(q)
    ws = torch.zeros(lib.afx_attention_ws_bytes(1, H, S), dtype=torch.uint8, device='cuda')
    junk = torch.empty(64 << 20, dtype=torch.float32, device='cuda')
    ref_o = ref_ws = None
    bad_o = bad_ws = 0
    rows_bad = set()
    for i in range(200):
        if i % 3 == 0:
            junk.normal_()
        _lib.check(lib.afx_attention_bf16(p(q), H * 128, p(k), H * 128, p(v), H * 128, p(o), H * 128, p(ws), 1, H, S, st()))
        torch.cuda.synchronize()
        if ref_o is None:
            ref_o, ref_ws = o.clone(), ws.clone()
            continue
        if not torch.equal(ws, ref_ws):
            bad_ws += 1
        if not torch.equal(o, ref_o):
            bad_o += 1
            d = (o.float() - ref_o.float()).abs()
            rr = d.amax(dim=1).nonzero().flatten().tolist()
            cc = d.amax(dim=0).nonzero().flatten().tolist()
            rows_bad.add((len(rr), rr[0], rr[-1], len(cc), cc[0] // 128, cc[-1] // 128))
    print(f'S={S} H={H}: V^T differs in {bad_ws} launches, O differs in {bad_o}; (n_rows, first, last, n_cols, head_first, head_last) = {sorted(rows_bad)[:8]}', flush=True)
