"""Gradient path through the MMDiT trunk for the rank-r LoRA adapters (reference: peft LoRA on the MLP linears,
lakonlab/models/architecture/arcflow/arcflux.py:294-302 with the target lists of configs/flux/arcflux_2nfe_k16.py:40-50
and configs/qwen/arcqwen_2nfe_k16.py:47-58; gradient checkpointing of every block arcflux.py:181-189).

Design for MI355X:
  * peft's forward  y = W x + b + B (A dropout(x))  is evaluated UNMERGED but as ONE GEMM per adapted linear: the rank-r product
    t = dropout(x) A^T is written into r extra columns of the activation buffer and the frozen weight carries B in r extra columns,
    y = [x | t] [W | B]^T  (K-extension; "wcat").  W stays the frozen tensor's values (nothing is re-merged after an optimizer
    step: only the B columns are refreshed), and small updates of B A are not lost in W's bf16 grid.
  * The training forward KEEPS every block's GEMM / attention outputs in HBM (the stash: 54 GB for FLUX at 4 samples; the reference
    checkpoints and recomputes every block, arcflux.py:181-189, to fit 80 GB).  BACKWARD walks the blocks in reverse, reads them back, redoes
    the element-wise work only (LayerNorm-modulate, RMSNorm + RoPE, GELU, dropout masks) and differentiates by hand (ARCFLOW_TRAIN_RECOMPUTE=1:
    each block is recomputed from its checkpoint with the same kernels instead): the input gradient and dT = dy B come out of ONE GEMM on the transposed frozen weight with r extra rows
    [dx0 | dT] = dy [W^T ; B^T]^T  (N-extension; "wtcat"), then dx = dx0 + ((dT A) . keep/(1-p));
    flash-attention backward, LN / RMSNorm+RoPE / GELU backward kernels.
  * LoRA gradients per adapted linear:   dB += dy^T t,  dA += dT^T dropout(x)   (rank-r products over the tokens, fp32 accumulate-into; round 5:
    ONE TN launch each on the token-major operands -- afx_tn.hip gathers the MFMA fragments with ds_read_b64_tr_b16 -- instead of two transposes + an NT launch).
  * LoRA input dropout (0.05 in the reference config): a counter-hash mask regenerated wherever it is needed (forward,
    recompute, backward) from (step seed, adapter, global token row, column); nothing is stored across the forward.
  * The timestep-embedder LoRA pair is trained too: its gradient is the sum of EVERY block's AdaLN modulation gradients
    (d_shift / d_scale / d_gate, accumulated per sample into a [n_mod] vector during the block backwards) pulled back
    through the stacked modulation matrix; the two [B <= 4]-row linears run on the gemv kernels.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import math
from typing import Dict, List, Optional, Tuple

import torch

from .. import _lib, ops
from ..engine import MMDiTEngine


_LORA_SPLITK = os.environ.get('ARCFLOW_LORA_SPLITK', '1') != '0'      # (0: A/B runs)
_LORA_DROPRES = os.environ.get('ARCFLOW_LORA_DROPRES', '1') != '0'    # the LoRA branch's input gradient masked + added in the GEMM epilogue (0: separate pass, A/B)
_LORA_TN = os.environ.get('ARCFLOW_LORA_TN', '1') != '0'              # weight gradients by the TN kernel (0: transposes + NT kernel, A/B runs)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class LoraSpec:
    """One adapted linear: where its weight lives in the packed set and where its A / B live in the flat buffers."""

    def __init__(self, name: str, packed_key: str, row0: int, out_f: int, in_f: int, off_a: int, off_b: int):
        self.name, self.packed_key, self.row0, self.out_f, self.in_f = name, packed_key, row0, out_f, in_f
        self.off_a, self.off_b = off_a, off_b


def lora_targets(family: str, num_double: int, num_single: int, D: int) -> List[Tuple[str, str, int, int, int]]:
    """(peft module name, packed weight key, first row inside that packed weight, out, in) of every adapted linear."""
    t = []
    for i in range(num_double):
        for s, ff in (('img', 'ff' if family == 'flux' else 'img_mlp'), ('txt', 'ff_context' if family == 'flux' else 'txt_mlp')):
            if family == 'qwen' and s == 'txt' and i == num_double - 1:
                continue                                   # txt_mlp of the last Qwen block is not adapted (range(59))
            t.append((f'transformer_blocks.{i}.{ff}.net.0.proj', f'd{i}.{s}_mlp1', 0, 4 * D, D))
            t.append((f'transformer_blocks.{i}.{ff}.net.2', f'd{i}.{s}_mlp2', 0, D, 4 * D))
    for i in range(num_single):
        t.append((f'single_transformer_blocks.{i}.proj_mlp', f's{i}.fused', 3 * D, 4 * D, D))
        t.append((f'single_transformer_blocks.{i}.proj_out', f's{i}.out', 0, D, 5 * D))
    # 'timestep_embedder.linear_1/2' of lora_target_modules (arcflux_2nfe_k16.py:46-47, arcqwen_2nfe_k16.py:50-51)
    t.append(('time_text_embed.timestep_embedder.linear_1', 'temb.t.l1', 0, D, 256))
    t.append(('time_text_embed.timestep_embedder.linear_2', 'temb.t.l2', 0, D, D))
    return t


class LoraTrunk:
    def __init__(self, student: MMDiTEngine, packed: Dict[str, torch.Tensor], rank: int, params: torch.Tensor,
                 base_offset: int, generator: Optional[torch.Generator] = None):
        """packed: the student's fused weight dict (frozen tensors shared with the teacher).  Adapted weights are
        replaced IN THIS DICT by private merged copies.  params: the distiller's flat fp32 parameter buffer; the LoRA
        A/B matrices are laid out from ``base_offset`` on (use ``LoraTrunk.num_params`` to size it)."""
        self.eng, self.packed, self.r = student, packed, rank
        self.family, self.D, self.H = student.family, student.dim, student.heads
        self.nd, self.ns = student.num_double, student.num_single
        self.dev = student.device
        D = self.D
        self.specs: List[LoraSpec] = []
        off = base_offset
        for name, key, row0, out_f, in_f in lora_targets(self.family, self.nd, self.ns, D):
            self.specs.append(LoraSpec(name, key, row0, out_f, in_f, off, off + rank * in_f))
            off += rank * in_f + out_f * rank
        self.end_offset = off
        self.params = params
        self.by_key: Dict[str, List[LoraSpec]] = {}
        for sp in self.specs:
            self.by_key.setdefault(sp.packed_key, []).append(sp)
        # adapted block linears: the frozen weight with r (padded to the GEMM's K granule) extra columns for B, and its transpose
        # with as many extra rows for B^T.  The frozen tensors in `packed` stay shared with the teacher and are never written.
        self.rp = (rank + 63) // 64 * 64
        self.base: Dict[str, torch.Tensor] = {}          # the timestep-embedder pair's frozen weights (gemv path)
        self.wcat: Dict[str, torch.Tensor] = {}          # key -> [out_total, in + rp] bf16 = [W | B | 0]
        self.wtcat: Dict[str, torch.Tensor] = {}         # key -> [in + rp, out_total] bf16 = [W^T ; B^T ; 0]
        for key in self.by_key:
            w = packed[key + '.weight']
            if key.startswith('temb.'):
                self.base[key] = w
                continue
            o, i = w.shape
            wc = torch.zeros(o, i + self.rp, dtype=torch.bfloat16, device=self.dev)
            wc[:, :i].copy_(w)
            wt = torch.zeros(i + self.rp, o, dtype=torch.bfloat16, device=self.dev)
            wt[:i].copy_(ops.transpose(w))
            self.wcat[key], self.wtcat[key] = wc, wt
        # peft init_lora_weights='gaussian': A ~ N(0, 1/r), B = 0
        for sp in self.specs:
            a = torch.randn(rank, sp.in_f, generator=generator, device=self.dev if generator is None or generator.device.type == 'cuda' else 'cpu') / rank
            self.A(sp).copy_(a.to(self.dev))
            self.B(sp).zero_()
        self.wt: Dict[str, torch.Tensor] = {}        # transposed FROZEN weights of the un-adapted linears (dgrad GEMMs)
        self.a16: Dict[str, torch.Tensor] = {}
        self.b16: Dict[str, torch.Tensor] = {}
        self.at16: Dict[str, torch.Tensor] = {}
        self.a16p: Dict[str, torch.Tensor] = {}      # A padded to rp rows ([rp, in]) and its transpose ([in, rp]) for the block GEMMs
        self.at16p: Dict[str, torch.Tensor] = {}
        self.bt16: Dict[str, torch.Tensor] = {}
        self.ybuf: Optional[torch.Tensor] = None     # [2 nd + ns, B*S, D] pre-gate branch outputs of the last training forward
        self.dmod: Optional[torch.Tensor] = None     # [n_mod] fp32: modulation gradients of the sample being back-propagated
        self.p_drop = 0.0           # LoRA input dropout (peft lora_dropout); masks are regenerated from (seed, site, row, col)
        self.seed = 0
        self.row0 = 0               # global row of this sample's first token: the batched masks are indexed by global row
        self._ones: Dict[int, torch.Tensor] = {}
        self._merged: Dict[str, torch.Tensor] = {}   # bind_merged(): private merged copies for the student engine, on request
        self._merged_dirty = True                    # the engine's weights are NOT W + B A of the live adapters (refresh() sets it, bind_merged() clears it)
        # Keep instead of recompute (288 GB HBM): the training forward leaves the outputs of every block's GEMMs and attention in `stash`
        # and the backward reads them back; only element-wise work (LayerNorm-modulate, RoPE, GELU, dropout) is redone.  The reference
        # checkpoints every block and recomputes its whole forward (arcflux.py:181-189) because it has to fit 80 GB.  ARCFLOW_TRAIN_RECOMPUTE=1
        # (or use_stash = False) selects the recompute path (A/B runs, the parity tests run both).
        self.use_stash = os.environ.get('ARCFLOW_TRAIN_RECOMPUTE', '0') != '1'
        # ... and (round 5) the adapters' dropped inputs xd = dropout(x): the backward needs them for dA += dT^T xd only, and regenerating them cost a
        # LayerNorm-modulate / GELU pass plus a dropout pass per adapted linear (37 GB more for FLUX at 4 samples; ARCFLOW_TRAIN_KEEP_XD=0: regenerate, A/B)
        self.keep_xd = os.environ.get('ARCFLOW_TRAIN_KEEP_XD', '1') != '0'
        # The text stream of a double block (512 / <= 128 rows beside 4096) is a chain of launches that fill a fraction of the chip; between
        # the two attention joins it is independent of the image stream, so it runs on a SIDE HIP stream and its work-groups take the
        # compute units the image stream's launches leave idle in their last rounds (ARCFLOW_TRAIN_TXT_STREAM=0: one stream, for A/B runs).
        self.side = torch.cuda.Stream(device=self.dev) if os.environ.get('ARCFLOW_TRAIN_TXT_STREAM', '1') != '0' else None
        # The adapters' weight gradients (dB += dy^T t, dA += dT^T dropout(x): two transposes and a [r x .] product each) are needed by
        # nobody until the block's slice is exchanged, and their launches have 12-60 tiles: they go to a THIRD stream behind the dgrad
        # GEMM that produced dT and run beside the input-gradient chain (ARCFLOW_TRAIN_WGRAD_STREAM=0: in line).
        self.aux = torch.cuda.Stream(device=self.dev) if os.environ.get('ARCFLOW_TRAIN_WGRAD_STREAM', '1') != '0' else None
        self.stash: Optional[Dict[str, torch.Tensor]] = None
        self._lse: Dict[Tuple[int, int], torch.Tensor] = {}
        self.fp8 = False            # block linears' forward / recompute on the fp8 MFMA (enable_fp8)
        self.wq: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}     # packed key -> (e4m3 [out, in], row scales fp32 [out]) of the FROZEN weight
        self._build_frozen_transposes()
        self.refresh()

    @staticmethod
    def num_params(family: str, nd: int, ns: int, D: int, rank: int) -> int:
        return sum(rank * i + o * rank for _, _, _, o, i in lora_targets(family, nd, ns, D))

    def A(self, sp: LoraSpec, flat: Optional[torch.Tensor] = None) -> torch.Tensor:
        f = self.params if flat is None else flat
        return f[sp.off_a:sp.off_a + self.r * sp.in_f].view(self.r, sp.in_f)

    def B(self, sp: LoraSpec, flat: Optional[torch.Tensor] = None) -> torch.Tensor:
        f = self.params if flat is None else flat
        return f[sp.off_b:sp.off_b + sp.out_f * self.r].view(sp.out_f, self.r)

    # ------------------------------------------------------------------ weights
    def _build_frozen_transposes(self):
        for i in range(self.nd):
            for s in ('img', 'txt'):
                for nm in ('qkv', 'out', 'mlp1', 'mlp2'):
                    k = f'd{i}.{s}_{nm}'
                    if k not in self.by_key:                     # adapted linears use wtcat (frozen W^T + the B^T rows)
                        self.wt[k] = ops.transpose(self.packed[k + '.weight'])

    def refresh(self):
        """After an optimizer step (or a checkpoint load): bf16 working copies of A / B and the B columns / B^T rows of the extended
        weights.  Nothing else changes -- W is frozen.  Whatever bind_merged() gave the student engine is stale from here on."""
        r = self.r
        self._merged_dirty = True
        for sp in self.specs:
            a16 = ops.cast_bf16(self.A(sp))
            b16 = ops.cast_bf16(self.B(sp))
            self.a16[sp.name], self.b16[sp.name] = a16, b16
            self.bt16[sp.name] = ops.transpose(b16)              # [r, out]
            self.at16[sp.name] = ops.transpose(a16)              # [in, r]
            if sp.packed_key in self.wcat:
                if self.rp == r:
                    self.a16p[sp.name], self.at16p[sp.name] = a16, self.at16[sp.name]
                else:                                            # rank below the GEMM's K granule: zero rows / columns up to rp
                    ap = torch.zeros(self.rp, sp.in_f, dtype=torch.bfloat16, device=self.dev)
                    ap[:r] = a16
                    self.a16p[sp.name], self.at16p[sp.name] = ap, ops.transpose(ap)
                self.wcat[sp.packed_key][sp.row0:sp.row0 + sp.out_f, sp.in_f:sp.in_f + r].copy_(b16)
                self.wtcat[sp.packed_key][sp.in_f:sp.in_f + r, sp.row0:sp.row0 + sp.out_f].copy_(self.bt16[sp.name])

    # ------------------------------------------------------------------ fp8 forward (BASELINE.json configs[4]: "fp8 MFMA fwd + bf16 grads")
    def _linear_keys(self) -> List[str]:
        keys = [f'd{i}.{s}_{n}' for i in range(self.nd) for s in ('img', 'txt') for n in ('qkv', 'out', 'mlp1', 'mlp2')]
        return keys + [f's{i}.{n}' for i in range(self.ns) for n in ('fused', 'out')]

    def enable_fp8(self, shared: Optional[Dict[str, torch.Tensor]] = None) -> None:
        """Run every block linear of the student's FORWARD (and of the backward's recompute, which must reproduce it) as
        y = s_a[m] s_w[n] (e4m3(x) . e4m3(W)^T) on the fp8 MFMA: activations quantised per token right before each GEMM, the FROZEN
        weights per output row, once; the LoRA branch B (A dropout(x)) stays a bf16 rank-r product added to it.  The BACKWARD is
        unchanged: dgrad on the bf16 W^T, LoRA gradients from the bf16 activations (straight-through: the quantiser has no
        gradient of its own).  shared: a weight dict that already holds '<key>.weight_q' / '<key>.wscale' of the frozen linears
        (the teacher engine after enable_fp8()) -- reused instead of a second copy."""
        self.fp8 = True
        for key in self._linear_keys():
            if shared is None or key + '.weight_q' not in shared:
                self.wq[key] = ops.quant_rows_fp8(self.packed[key + '.weight'])
            else:
                self.wq[key] = (shared[key + '.weight_q'], shared[key + '.wscale'])

    def _lin(self, x: torch.Tensor, key: str, out: Optional[torch.Tensor] = None, **kw) -> torch.Tensor:
        """An UN-adapted block linear: x [M, K] @ packed[key].weight.T + bias through the bf16 or the fp8 GEMM."""
        w, b = self.packed[key + '.weight'], self.packed[key + '.bias']
        if not self.fp8:
            return ops.linear(x, w, b, out=out, **kw)
        q, sc = self.wq[key]
        aq, asc = ops.quant_rows_fp8(x)
        return ops.linear_fp8(aq, asc, q, sc, b, out=out, **kw)

    def _xe(self, rows: int, in_f: int) -> torch.Tensor:
        """[rows, in_f + rp] activation buffer of an adapted linear: x goes into [:, :in_f], t = dropout(x) A^T into the next rp
        columns (A is zero-padded to rp rows, so the columns past r come out as exact zeros)."""
        return torch.empty(rows, in_f + self.rp, dtype=torch.bfloat16, device=self.dev)

    def _adapted(self, sp: LoraSpec, xe: torch.Tensor, row_off: int, out: Optional[torch.Tensor] = None, main: bool = True,
                 xd_out: Optional[torch.Tensor] = None):
        """peft LoRA linear on xe = [x | . ] ([M, in + rp], x already in place):  xd = dropout(x);  t = xd A^T -> xe[:, in:in+r];
        out = [x | t] [W | B]^T + bias for ALL rows of the packed weight (the fused single-block weight carries zero B columns on its
        k|v|q rows).  main=False: only xd and t (the recompute of a branch's last linear, whose output the backward does not need).
        xd_out: where dropout(x) is written (the stash).  Returns (xd, t, out)."""
        i, r, key = sp.in_f, self.r, sp.packed_key
        x, t = xe[:, :i], xe[:, i:i + r]
        if self.p_drop <= 0:
            xd = x
            if xd_out is not None:
                xd_out.copy_(x)             # (kept for the backward's dA product: x itself lives in a temporary operand buffer)
        else:                               # xd_out: straight into the stash
            xd = ops.lora_dropout(x, self.p_drop, self._site_seed(sp), self.row0 + row_off, mode=1, out=xd_out)
        self._skinny(xd, self.a16p[sp.name], out=xe[:, i:])
        if not main:
            return xd, t, None
        b = self.packed[key + '.bias']
        if not self.fp8:
            return xd, t, ops.linear(xe, self.wcat[key], b, out=out)
        q, sc = self.wq[key]
        aq, asc = ops.quant_rows_fp8(x)
        y = ops.linear_fp8(aq, asc, q, sc, b, out=out)
        yl = y[:, sp.row0:sp.row0 + sp.out_f]
        # + t B^T (K = rp), added in place by the product's own epilogue (plain residual add: no [M, out] correction in memory, no add pass)
        ops.linear(xe[:, i:], self.wcat[key][sp.row0:sp.row0 + sp.out_f, i:], epilogue='gate_res', residual=yl, out=yl)
        return xd, t, y

    def _dropped(self, sp: LoraSpec, x: torch.Tensor, row_off: int) -> torch.Tensor:
        """dropout(x) of adapter `sp` (x itself without dropout): the mask is regenerated from (step seed, adapter, global row, column)."""
        if self.p_drop <= 0:
            return x
        return ops.lora_dropout(x, self.p_drop, self._site_seed(sp), self.row0 + row_off, mode=1)

    def _adapted_backward(self, sp: LoraSpec, dy: torch.Tensor, xd: torch.Tensor, t: torch.Tensor, grads: torch.Tensor, row_off: int):
        """dy [M, out_total] (all rows of the packed weight), xd = dropout(x) [M, in], t = xd A^T [M, r] from the recompute.
        dB += dy^T t;  dA += (dy B)^T xd;  returns dx = dy W + ((dy B) A) . keep/(1-p)  as a view [M, in] of an [M, in + rp] buffer."""
        i, r = sp.in_f, self.r
        dxe = ops.linear(dy, self.wtcat[sp.packed_key])                       # [dx0 | dT | 0]
        dx, dT = dxe[:, :i], dxe[:, i:i + r]
        dyl = dy[:, sp.row0:sp.row0 + sp.out_f]
        cur = torch.cuda.current_stream(self.dev)
        if self.aux is not None:
            self.aux.wait_stream(cur)
            for x_ in (dy, t, xd, dxe):                    # the allocator must not hand their memory out again before the side work has read it
                x_.record_stream(self.aux)
        with (torch.cuda.stream(self.aux) if self.aux is not None else contextlib.nullcontext()):
            gB, gA = self.B(sp, grads), self.A(sp, grads)
            if _LORA_TN and all(x_.data_ptr() % 16 == 0 for x_ in (dyl, t, dT, xd, gB, gA)):
                # contraction over the tokens straight from the token-major operands (afx_tn.hip: LDS transpose reads; no transposed copies)
                ops.linear_tn_f32out(dyl, t, out=gB, accumulate=True)
                ops.linear_tn_f32out(dT, xd, out=gA, accumulate=True)
            else:               # (ARCFLOW_LORA_TN=0, A/B -- or a slice of the flat buffers that is not 16-byte aligned: the round-4 path, four transposes + the NT kernel)
                ops.linear_f32out(ops.transpose(dyl, 64), ops.transpose(t, 64), out=self.B(sp, grads), accumulate=True)
                ops.linear_f32out(ops.transpose(dT, 64), ops.transpose(xd, 64), out=self.A(sp, grads), accumulate=True)
        if _LORA_DROPRES:   # dx = dx0 + ((dy B) A) . keep/(1-p): the mask is applied in the product's epilogue (round 5)
            ops.linear_dropres(dxe[:, i:], self.at16p[sp.name], dx, self.p_drop, self._site_seed(sp), self.row0 + row_off, out=dx)
        else:               # (ARCFLOW_LORA_DROPRES=0, A/B: the product to memory, then a mask-and-add pass)
            ops.lora_dropout(ops.linear(dxe[:, i:], self.at16p[sp.name]), self.p_drop, self._site_seed(sp), self.row0 + row_off, mode=3, out=dx)
        return dx

    def block_slice(self, block: int) -> Tuple[int, int]:
        """[a, b) of the flat parameter / gradient buffer holding the adapters of transformer block ``block``
        (0 .. nd-1 double, nd .. nd+ns-1 single); the layout is block-major, so the range is contiguous."""
        pre = f'd{block}.' if block < self.nd else f's{block - self.nd}.'
        sps = [sp for sp in self.specs if sp.packed_key.startswith(pre)]
        if not sps:
            return (0, 0)
        return (min(sp.off_a for sp in sps), max(sp.off_b + sp.out_f * self.r for sp in sps))

    def merged_state(self) -> Dict[str, torch.Tensor]:
        """W + B A per adapted linear (fp32 accumulate, one rounding) -- built on request (export, tests); training never merges."""
        out = {}
        for sp in self.specs:
            w = self.packed[sp.packed_key + '.weight'][sp.row0:sp.row0 + sp.out_f]
            ones = self._ones.setdefault(sp.in_f, torch.ones(1, sp.in_f, dtype=torch.float32, device=self.dev))
            if self.r % 64 == 0:
                b16, at = self.b16[sp.name], self.at16[sp.name]
            else:
                b16 = torch.zeros(sp.out_f, self.rp, dtype=torch.bfloat16, device=self.dev)
                b16[:, :self.r] = self.b16[sp.name]
                at = torch.zeros(sp.in_f, self.rp, dtype=torch.bfloat16, device=self.dev)
                at[:, :self.r] = self.at16[sp.name]
            out[sp.name] = ops.linear(b16, at, None, epilogue='gate_res', gate=ones, residual=w.contiguous(), rows_per_batch=sp.out_f)
        return out

    def ensure_merged(self) -> None:
        """bind_merged() if the adapters changed since the last one (every optimizer step / checkpoint load does): what a forward of the student
        ENGINE has to call first -- bound to the frozen weights it would silently evaluate the un-adapted trunk with the trained heads."""
        if self._merged_dirty:
            self.bind_merged()

    def bind_merged(self) -> None:
        """Give the student ENGINE the merged weights W + B A of the live adapters (validation / inference with the distiller's own
        student; the training step itself never needs them: its blocks run through this trunk on the extended frozen weights).
        ArcFlowDistiller.student_forward() does this lazily through ensure_merged()."""
        ms = self.merged_state()
        upd = {}
        for key, sps in self.by_key.items():
            m = self._merged.get(key)
            if m is None:
                m = self._merged[key] = self.packed[key + '.weight'].clone()
            for sp in sps:
                m[sp.row0:sp.row0 + sp.out_f].copy_(ms[sp.name])
            upd[key + '.weight'] = m
        self.eng.bind_packed(upd)
        self._merged_dirty = False

    # ------------------------------------------------------------------ LoRA gradients of one linear
    def _dmod_ln(self, x: torch.Tensor, dxn: torch.Tensor, off_scale: int, off_shift: int) -> None:
        """xn = LN(x) (1 + scale) + shift:  d_scale += sum_rows dxn * LN(x),  d_shift += sum_rows dxn."""
        # straight into the sample's modulation-gradient vector (float atomics per column: the two token streams add into different slices)
        ops.normout_backward_split(x, dxn, self.dmod[off_scale:off_scale + self.D], self.dmod[off_shift:off_shift + self.D])

    # ------------------------------------------------------------------ timestep embedder with its LoRA pair (host-sized math)
    def _keep_scale(self, sp: LoraSpec, rows: int, cols: int) -> Optional[torch.Tensor]:
        if self.p_drop <= 0:
            return None
        ones = torch.ones(rows, cols, dtype=torch.bfloat16, device=self.dev)
        keep = ops.lora_dropout(ones, self.p_drop, self._site_seed(sp), 0, mode=1) > 0
        return keep.float() / (1.0 - self.p_drop)

    def temb_forward(self, sigma: torch.Tensor) -> torch.Tensor:
        """timestep_embedder(sincos(1000 sigma)) [B, D] fp32 with the LoRA branches B A dropout(.) on both linears
        (diffusers Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0) -> Linear -> SiLU -> Linear).  The
        [B<=4]-row products run on the weight-streaming gemv kernel; the intermediates are kept for ``temb_backward``."""
        sp1, sp2 = self._spec('temb.t.l1'), self._spec('temb.t.l2')
        # the reference casts the timestep to the trunk dtype before the sinusoid, FLUX also the x1000 product
        # (arcflux.py:160-162, arcqwen.py:128; the engine's sincos kernel does the same)
        t = sigma.to(self.dev, torch.float32).reshape(-1).bfloat16().float() * 1000.0
        if self.family == 'flux':
            t = t.bfloat16().float()
        freqs = torch.exp(-math.log(10000.0) * torch.arange(128, dtype=torch.float32, device=self.dev) / 128.0)
        ang = t[:, None] * freqs[None]
        s = torch.cat([ang.cos(), ang.sin()], dim=1)                                   # [B, 256], cos first
        W1, W2 = self.base['temb.t.l1'], self.base['temb.t.l2']                      # frozen bf16 [out, in]
        b1, b2 = self.packed['temb.t.l1.bias'], self.packed['temb.t.l2.bias']
        k1, k2 = self._keep_scale(sp1, s.shape[0], 256), self._keep_scale(sp2, s.shape[0], self.D)
        sd = (s if k1 is None else s * k1).contiguous()
        t1 = ops.gemv(sd, self.a16[sp1.name])                                          # [B, r]  = dropout(s) A1^T
        u = ops.gemv(s.contiguous(), W1, b1)
        ops.gemv(t1, self.b16[sp1.name], out=u, accumulate=True)                       # + (.) B1^T
        h = torch.nn.functional.silu(u)
        hd = (h if k2 is None else h * k2).contiguous()
        t2 = ops.gemv(hd, self.a16[sp2.name])
        y = ops.gemv(h.contiguous(), W2, b2)
        ops.gemv(t2, self.b16[sp2.name], out=y, accumulate=True)
        self._temb_cache = (sd, u, hd, k2, t1, t2)
        return y.contiguous()

    def temb_backward(self, dtemb: torch.Tensor, grads: torch.Tensor) -> None:
        """dtemb [B, D] = d loss / d timestep-embedding output; accumulates dA, dB of both linears into ``grads``
        (gemv / transposed gemv / rank-B outer-product kernels: no torch matmul on the path)."""
        sp1, sp2 = self._spec('temb.t.l1'), self._spec('temb.t.l2')
        sd, u, hd, k2, t1, t2 = self._temb_cache
        dtemb = dtemb.contiguous()
        B = dtemb.shape[0]
        dT2 = ops.gemv(dtemb, self.bt16[sp2.name])                                     # [B, r] = dtemb B2
        ops.outer_accum(dtemb, t2, self.B(sp2, grads))                                 # dB2 += dtemb^T (hd A2^T)
        ops.outer_accum(dT2, hd, self.A(sp2, grads))                                   # dA2 += dT2^T hd
        dh = torch.zeros(B, self.D, dtype=torch.float32, device=self.dev)
        ops.gemv_t(dtemb, self.base['temb.t.l2'], dh)                                  # dtemb W2
        dl = torch.zeros(B, self.D, dtype=torch.float32, device=self.dev)
        ops.gemv_t(dT2, self.a16[sp2.name], dl)                                        # (dT2 A2) . keep_scale
        dh += dl if k2 is None else dl * k2
        sg = torch.sigmoid(u)
        du = (dh * (sg * (1 + u * (1 - sg)))).contiguous()
        ops.outer_accum(du, t1, self.B(sp1, grads))                                    # dB1 += du^T (sd A1^T)
        dT1 = ops.gemv(du, self.bt16[sp1.name])                                        # [B, r] = du B1
        ops.outer_accum(dT1, sd, self.A(sp1, grads))                                   # dA1 += dT1^T sd

    def _site_seed(self, sp: LoraSpec) -> int:
        return (self.seed * 0x9E3779B1 + (sp.off_a * 2654435761 % (1 << 32))) & 0xffffffff

    @staticmethod
    def _skinny(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [M, K] @ w[r, K].T with r = the LoRA rank (256 columns): a launch of 8-72 tiles whose time is the length of its K loop.  For the
        K >= 8192 sites (mlp2's / proj_out's x A^T) eight K chunks into fp32 slabs + one fold pass halve it (101 -> 49 us at
        4096 x 256 x 12288, 98 -> 34 us for the 512-row text stream; at K = 3072 the plain launch is as fast: tools/skinny_bench.py)."""
        if w.shape[1] >= 8192 and _LORA_SPLITK:
            return ops.linear_splitk(x, w, out=out, split_k=8 if x.shape[0] > 1024 else 16)
        return ops.linear(x, w, out=out)

    # ------------------------------------------------------------------ helpers on strided 2-D views
    def _rope(self, x, y, w_txt, w_img, cos, sin, S, T, dy=None):
        lib = self.eng.lib
        _lib.check(lib.afx_qk_norm_rope_oop_bf16(_p(x), x.stride(0), _p(y), y.stride(0), _p(dy), 0 if dy is None else dy.stride(0),
                                                 _p(w_txt), _p(w_img), _p(cos), _p(sin), 1, S, T, self.H, int(dy is not None), _s()))

    def _streams(self, T: int, S: int):
        """The two token streams of a double block as (name, rows, index, stream context): the text stream first, on the side stream --
        it waits for everything enqueued on the current stream so far, and ``_join`` makes the current stream wait for it.  Tensors a
        stream allocates inside its context are only used by that stream until the join (the caching allocator keeps them in that
        stream's pool); tensors shared by both are allocated before the loop and outlive the join."""
        if self.side is None or T == 0:
            return (('img', slice(T, S), 0, contextlib.nullcontext()), ('txt', slice(0, T), 1, contextlib.nullcontext()))
        self.side.wait_stream(torch.cuda.current_stream(self.dev))
        return (('txt', slice(0, T), 1, torch.cuda.stream(self.side)), ('img', slice(T, S), 0, contextlib.nullcontext()))

    def _join(self) -> None:
        if self.side is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)

    def _join_wgrad(self) -> None:
        """The current stream waits for the weight-gradient stream: before a block's gradient slice is handed on, before the stash rows
        it reads are overwritten."""
        if self.aux is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.aux)

    # ------------------------------------------------------------------ the stash of one student forward
    def stash_bytes(self, rows: int) -> int:
        D = self.D
        xd = (self.nd * 5 * D + self.ns * 6 * D) if self.keep_xd else 0
        return 2 * rows * (self.nd * 9 * D + self.ns * 8 * D + 2 * (self.nd + self.ns) * self.rp + xd)

    def _stash_for(self, rows: int) -> Optional[Dict[str, torch.Tensor]]:
        """Per (block, token): double blocks k|v|q pre-norm (3D), O (D), X1 (D), the mlp pre-activation (4D); single blocks the fused
        k|v|q|mlp pre-activation (7D) and O (D); t = dropout(x) A^T of both adapters of every block (2 rp); with keep_xd the adapters' dropped
        inputs (5D / 6D).  FLUX at 4 samples: 54 GB + 37 GB."""
        if self.stash is None or self.stash['rows'] < rows:      # grow-only: prompt lengths (Qwen-Image) vary from batch to batch, the row ranges used are [0, B S)
            D, bf = self.D, dict(dtype=torch.bfloat16, device=self.dev)
            self.stash = None
            self._lse.clear()
            free, _ = torch.cuda.mem_get_info(self.dev)
            room = free + torch.cuda.memory_reserved(self.dev) - torch.cuda.memory_allocated(self.dev) - (8 << 30)
            need = self.stash_bytes(rows)
            if need > room and self.keep_xd:
                # three levels, the cheapest one that fits: GEMM outputs + dropped adapter inputs > GEMM outputs only (round 4's stash) > recompute
                import warnings
                self.keep_xd = False
                warnings.warn(f'LoraTrunk: {need / 2**30:.0f} GiB of kept forward outputs do not fit ({room / 2**30:.0f} GiB available): keeping the GEMM / '
                              f'attention outputs only ({self.stash_bytes(rows) / 2**30:.0f} GiB; the adapters\' dropped inputs are recomputed in the backward)')
                need = self.stash_bytes(rows)
            if need > room:
                # not enough HBM for this batch / sequence: the reference's schedule (recompute every block from its checkpoint) still works
                import warnings
                warnings.warn(f'LoraTrunk: {need / 2**30:.0f} GiB of kept forward outputs do not fit ({room / 2**30:.0f} GiB available): '
                              f'falling back to recomputing every block in the backward')
                self.use_stash = False
                return None
            self.stash = dict(rows=rows, qkv=torch.empty(self.nd, rows, 3 * D, **bf), o=torch.empty(self.nd + self.ns, rows, D, **bf),
                              x1=torch.empty(self.nd, rows, D, **bf), pre=torch.empty(self.nd, rows, 4 * D, **bf),
                              fp=torch.empty(self.ns, rows, 7 * D, **bf), t=torch.empty(2 * (self.nd + self.ns), rows, self.rp, **bf))
            if self.keep_xd:      # dropout(x) of every adapted linear: double blocks mlp1 (D) / mlp2 (4D) inputs, single blocks proj_mlp (D) / proj_out (5D) inputs
                self.stash.update(xd1=torch.empty(self.nd, rows, D, **bf), xd2=torch.empty(self.nd, rows, 4 * D, **bf),
                                  xdm=torch.empty(self.ns, rows, D, **bf), xdo=torch.empty(self.ns, rows, 5 * D, **bf))
        return self.stash

    # ------------------------------------------------------------------ block forward / recompute + backward (one sample)
    def _double_block(self, i: int, X: torch.Tensor, mod: torch.Tensor, cos, sin, T: int, dXo: Optional[torch.Tensor], grads,
                      fwd_only: bool = False) -> torch.Tensor:
        """X [S, D] block input, mod [n_mod] f32 of this sample.  fwd_only: returns the block output (training forward with
        LoRA dropout).  Otherwise dXo [S, D] is the grad of the block output: recompute, backward, returns the grad of X."""
        D, S = self.D, X.shape[0]
        dev = self.dev
        pk, p = self.packed, f'd{i}.'
        m0 = i * 12 * D
        mv = {('img', c): mod[m0 + c * D:m0 + (c + 1) * D] for c in range(6)}
        mv.update({('txt', c): mod[m0 + (6 + c) * D:m0 + (7 + c) * D] for c in range(6)})
        qkn = pk[p + 'qknorm']                                   # [img_q, img_k, txt_q, txt_k]
        bf = dict(dtype=torch.bfloat16, device=dev)
        R = slice(self.row0, self.row0 + S)
        st = self.stash if self.use_stash else None
        restore = st is not None and not fwd_only           # backward with the forward's GEMM / attention outputs at hand
        keepf = st is not None and fwd_only                 # forward that leaves them there
        kxd = st is not None and 'xd1' in st                # ... and the adapters' dropped inputs
        QKVp = st['qkv'][i, R] if st is not None else torch.empty(S, 3 * D, **bf)       # pre-norm k | v | q
        if not restore:
            Xn1 = torch.empty(S, D, **bf)
            for s, rows, _, on_stream in self._streams(T, S):
                with on_stream:
                    ops.norm_modulate(X[rows], mv[(s, 1)], mv[(s, 0)], out=Xn1[rows])
                    self._lin(Xn1[rows], p + s + '_qkv', out=QKVp[rows])
            self._join()
        Kp, V, Qp = QKVp[:, :D], QKVp[:, D:2 * D], QKVp[:, 2 * D:]
        K, Q = torch.empty(S, D, **bf), torch.empty(S, D, **bf)
        O = st['o'][i, R] if st is not None else torch.empty(S, D, **bf)
        self._rope(Kp, K, qkn[3], qkn[1], cos, sin, S, T)
        self._rope(Qp, Q, qkn[2], qkn[0], cos, sin, S, T)
        if restore:
            lse = self._lse[(i, self.row0)]
        else:
            lse = ops.attention_fwd_lse_2d(Q, K, V, O, 1, S, self.H)
            if keepf:
                self._lse[(i, self.row0)] = lse
        X1 = st['x1'][i, R] if st is not None else torch.empty(S, D, **bf)
        Xe2 = self._xe(S, D)                                     # [Xn2 | t1]: the mlp1 operand
        He = self._xe(S, 4 * D)                                  # [gelu(Pre) | t2]: the mlp2 operand
        Xn2, Hh = Xe2[:, :D], He[:, :4 * D]
        Pre = st['pre'][i, R] if st is not None else torch.empty(S, 4 * D, **bf)
        Xo = torch.empty(S, D, **bf) if fwd_only else None
        keep = {}                                                # per stream: (xd1, t1, xd2, t2) of the two adapters
        for s, rows, _, on_stream in self._streams(T, S):
            with on_stream:
                sp1, sp2 = self._spec(p + s + '_mlp1'), self._spec(p + s + '_mlp2')
                if restore:         # t1 / t2 come back from the stash, and so do the dropped inputs (keep_xd) -- else element-wise work only: Xn2 -> dropout, gelu(Pre) -> dropout
                    xd1 = t1 = xd2 = t2 = None
                    if sp1 is not None:
                        if kxd:
                            xd1 = st['xd1'][i, R][rows]
                        else:
                            ops.norm_modulate(X1[rows], mv[(s, 4)], mv[(s, 3)], out=Xn2[rows])
                            xd1 = self._dropped(sp1, Xn2[rows], rows.start)
                        t1 = st['t'][2 * i, R][rows][:, :self.r]
                    if sp2 is not None:
                        if kxd:
                            xd2 = st['xd2'][i, R][rows]
                        else:
                            ops.gelu(Pre[rows], out=Hh[rows])
                            xd2 = self._dropped(sp2, Hh[rows], rows.start)
                        t2 = st['t'][2 * i + 1, R][rows][:, :self.r]
                    keep[s] = (xd1, t1, xd2, t2)
                    continue
                if fwd_only:        # keep the pre-gate branch output: the backward needs it for d_gate
                    y1 = self.ybuf[2 * i, self.row0:self.row0 + S][rows]
                    self._lin(O[rows], p + s + '_out', out=y1)
                    ops.gate_residual(y1, mv[(s, 2)], X[rows], out=X1[rows])
                else:
                    self._lin(O[rows], p + s + '_out', epilogue='gate_res', gate=mv[(s, 2)], residual=X[rows], out=X1[rows])
                ops.norm_modulate(X1[rows], mv[(s, 4)], mv[(s, 3)], out=Xn2[rows])
                if sp1 is not None:
                    xd1, t1, _ = self._adapted(sp1, Xe2[rows], rows.start, out=Pre[rows], xd_out=st['xd1'][i, R][rows] if (keepf and kxd) else None)
                    if keepf:
                        st['t'][2 * i, R][rows].copy_(Xe2[rows][:, D:])
                else:
                    xd1 = t1 = None
                    self._lin(Xn2[rows], p + s + '_mlp1', out=Pre[rows])
                ops.gelu(Pre[rows], out=Hh[rows])
                y2 = self.ybuf[2 * i + 1, self.row0:self.row0 + S][rows] if fwd_only else None
                if sp2 is not None:
                    xd2, t2, _ = self._adapted(sp2, He[rows], rows.start, out=y2, main=fwd_only, xd_out=st['xd2'][i, R][rows] if (keepf and kxd) else None)
                    if keepf:
                        st['t'][2 * i + 1, R][rows].copy_(He[rows][:, 4 * D:])
                else:
                    xd2 = t2 = None
                    if fwd_only:
                        self._lin(Hh[rows], p + s + '_mlp2', out=y2)
                if fwd_only:
                    ops.gate_residual(y2, mv[(s, 5)], X1[rows], out=Xo[rows])
                keep[s] = (xd1, t1, xd2, t2)
        self._join()
        if fwd_only:
            return Xo
        # ---- backward ----
        dX1 = torch.empty(S, D, **bf)
        dO = torch.empty(S, D, **bf)
        for s, rows, _, on_stream in self._streams(T, S):
            with on_stream:
                dY2 = ops.add_scale(dXo[rows], gate=mv[(s, 5)])
                sp1, sp2 = self._spec(p + s + '_mlp1'), self._spec(p + s + '_mlp2')
                xd1, t1, xd2, t2 = keep[s]
                if sp2 is not None:
                    dH = self._adapted_backward(sp2, dY2, xd2, t2, grads, rows.start)
                else:
                    dH = ops.linear(dY2, self.wt[p + s + '_mlp2'])
                dPre = ops.gelu(Pre[rows], dh=dH)
                if sp1 is not None:
                    dXn2 = self._adapted_backward(sp1, dPre, xd1, t1, grads, rows.start)
                else:
                    dXn2 = ops.linear(dPre, self.wt[p + s + '_mlp1'])
                ops.ln_modulate_backward(X1[rows], dXn2, mv[(s, 4)], dres=dXo[rows], out=dX1[rows])
                if self.dmod is not None:      # d(shift3, scale4, gate5, gate2) of this stream
                    so = m0 + (0 if s == 'img' else 6) * D
                    ops.coldot(dXo[rows], self.ybuf[2 * i + 1, self.row0:self.row0 + S][rows], self.dmod[so + 5 * D:so + 6 * D])
                    self._dmod_ln(X1[rows], dXn2, so + 4 * D, so + 3 * D)
                    ops.coldot(dX1[rows], self.ybuf[2 * i, self.row0:self.row0 + S][rows], self.dmod[so + 2 * D:so + 3 * D])
                dYo = ops.add_scale(dX1[rows], gate=mv[(s, 2)])
                ops.linear(dYo, self.wt[p + s + '_out'], out=dO[rows])
        self._join()
        dQ, dK = torch.empty(S, D, **bf), torch.empty(S, D, **bf)
        dQKVp = torch.empty(S, 3 * D, **bf)
        ops.attention_bwd_2d(Q, K, V, O, dO, lse, dQ, dK, dQKVp[:, D:2 * D], 1, S, self.H)
        self._rope(Kp, dQKVp[:, :D], qkn[3], qkn[1], cos, sin, S, T, dy=dK)
        self._rope(Qp, dQKVp[:, 2 * D:], qkn[2], qkn[0], cos, sin, S, T, dy=dQ)
        dX = torch.empty(S, D, **bf)
        for s, rows, _, on_stream in self._streams(T, S):
            with on_stream:
                dXn1 = ops.linear(dQKVp[rows], self.wt[p + s + '_qkv'])
                ops.ln_modulate_backward(X[rows], dXn1, mv[(s, 1)], dres=dX1[rows], out=dX[rows])
                if self.dmod is not None:
                    so = m0 + (0 if s == 'img' else 6) * D
                    self._dmod_ln(X[rows], dXn1, so + D, so)
        self._join()
        return dX

    def _single_block(self, i: int, X: torch.Tensor, mod: torch.Tensor, cos, sin, T: int, dXo: Optional[torch.Tensor], grads,
                      fwd_only: bool = False) -> torch.Tensor:
        D, S = self.D, X.shape[0]
        dev = self.dev
        pk, p = self.packed, f's{i}.'
        m0 = self.nd * 12 * D + i * 3 * D
        sh, sc, gt = mod[m0:m0 + D], mod[m0 + D:m0 + 2 * D], mod[m0 + 2 * D:m0 + 3 * D]
        qkn = pk[p + 'qknorm']                                   # [q, k]
        bf = dict(dtype=torch.bfloat16, device=dev)
        sp_out, sp_mlp = self._spec(p + 'out'), self._spec(p + 'fused')
        R = slice(self.row0, self.row0 + S)
        st = self.stash if self.use_stash else None
        restore = st is not None and not fwd_only
        keepf = st is not None and fwd_only
        bi = self.nd + i                                         # block index in the o / t / lse stashes
        kxd = st is not None and 'xdm' in st
        Fp = st['fp'][i, R] if st is not None else torch.empty(S, 7 * D, **bf)          # pre-activation k|v|q|mlp
        if restore and kxd:
            xd_m, t_m = st['xdm'][i, R], st['t'][2 * bi, R][:, :self.r]
        else:
            Xe = self._xe(S, D)                                  # [Xn | t_mlp]: operand of the fused k|v|q|mlp launch
            Xn = Xe[:, :D]
            ops.norm_modulate(X, sc, sh, out=Xn)
        if restore and not kxd:
            xd_m, t_m = self._dropped(sp_mlp, Xn, 0), st['t'][2 * bi, R][:, :self.r]
        elif not restore:
            xd_m, t_m, _ = self._adapted(sp_mlp, Xe, 0, out=Fp, xd_out=st['xdm'][i, R] if (keepf and kxd) else None)  # (the k|v|q rows of wcat carry zero B columns)
            if keepf:
                st['t'][2 * bi, R].copy_(Xe[:, D:])
        Kp, V, Qp, Mp = Fp[:, :D], Fp[:, D:2 * D], Fp[:, 2 * D:3 * D], Fp[:, 3 * D:]
        K, Q = torch.empty(S, D, **bf), torch.empty(S, D, **bf)
        self._rope(Kp, K, qkn[1], qkn[1], cos, sin, S, T)
        self._rope(Qp, Q, qkn[0], qkn[0], cos, sin, S, T)
        y = self.ybuf[2 * self.nd + i, self.row0:self.row0 + S] if fwd_only else None
        if restore and kxd:         # O, t and dropout([O | gelu(mlp)]) all come back from the stash: no GELU pass, no copy, no dropout pass
            lse = self._lse[(bi, self.row0)]
            O = st['o'][bi, R]
            xd_o, t_o = st['xdo'][i, R], st['t'][2 * bi + 1, R][:, :self.r]
        else:
            Ge = self._xe(S, 5 * D)                              # [O | gelu(mlp) | t_out] = proj_out operand
            G = Ge[:, :5 * D]
            O = G[:, :D]
            if restore:
                lse = self._lse[(bi, self.row0)]
                G[:, :D].copy_(st['o'][bi, R])
            else:
                lse = ops.attention_fwd_lse_2d(Q, K, V, G[:, :D], 1, S, self.H)
                if keepf:
                    self._lse[(bi, self.row0)] = lse
                    st['o'][bi, R].copy_(G[:, :D])
            ops.gelu(Mp, out=G[:, D:])
            if restore:
                xd_o, t_o = self._dropped(sp_out, G, 0), st['t'][2 * bi + 1, R][:, :self.r]
            else:
                xd_o, t_o, _ = self._adapted(sp_out, Ge, 0, out=y, main=fwd_only, xd_out=st['xdo'][i, R] if (keepf and kxd) else None)
            if keepf:
                st['t'][2 * bi + 1, R].copy_(Ge[:, 5 * D:])
        if fwd_only:
            return ops.gate_residual(y, gt, X)
        # ---- backward ----
        dY = ops.add_scale(dXo, gate=gt)
        dG = self._adapted_backward(sp_out, dY, xd_o, t_o, grads, 0)          # [S, 5D] (view)
        dFp = torch.empty(S, 7 * D, **bf)
        ops.gelu(Mp, dh=dG[:, D:], out=dFp[:, 3 * D:])
        dQ, dK = torch.empty(S, D, **bf), torch.empty(S, D, **bf)
        ops.attention_bwd_2d(Q, K, V, O, dG[:, :D], lse, dQ, dK, dFp[:, D:2 * D], 1, S, self.H)
        self._rope(Kp, dFp[:, :D], qkn[1], qkn[1], cos, sin, S, T, dy=dK)
        self._rope(Qp, dFp[:, 2 * D:3 * D], qkn[0], qkn[0], cos, sin, S, T, dy=dQ)
        dXn = self._adapted_backward(sp_mlp, dFp, xd_m, t_m, grads, 0)        # dgrad over all 7D columns; dT / dB from the mlp columns
        if self.dmod is not None:          # d(shift, scale, gate) of the single-stream AdaLN
            ops.coldot(dXo, self.ybuf[2 * self.nd + i, self.row0:self.row0 + S], self.dmod[m0 + 2 * D:m0 + 3 * D])
            self._dmod_ln(X, dXn, m0 + D, m0)
        return ops.ln_modulate_backward(X, dXn, sc, dres=dXo)

    def _spec(self, key: str) -> Optional[LoraSpec]:
        sps = self.by_key.get(key)
        return sps[0] if sps else None

    # ------------------------------------------------------------------ training forward of one sample (LoRA dropout active)
    def forward_sample(self, x_tokens: torch.Tensor, ckpt: torch.Tensor, b: int, mod_all: torch.Tensor, T: int, N: int, hp: int,
                       wp: int) -> None:
        """x_tokens [B*S, D]: the joint token matrix after the embedders (engine stage 1); sample b's rows are run through
        every block IN PLACE, each block input is stored in ckpt[block, b rows] for the backward."""
        S = T + N
        self.row0 = b * S
        mod = mod_all[b]
        cos, sin = self.eng.rope_tables(hp, wp, T)
        if self.ybuf is None or self.ybuf.shape[1] != x_tokens.shape[0]:
            self.ybuf = torch.empty(2 * self.nd + self.ns, x_tokens.shape[0], self.D, dtype=torch.bfloat16, device=self.dev)
        if self.use_stash:
            self._stash_for(x_tokens.shape[0])
        X = x_tokens[b * S:(b + 1) * S]
        for i in range(self.nd):
            ckpt[i, b * S:(b + 1) * S].copy_(X)
            X = self._double_block(i, X, mod, cos, sin, T, None, None, fwd_only=True)
        for i in range(self.ns):
            ckpt[self.nd + i, b * S:(b + 1) * S].copy_(X)
            X = self._single_block(i, X, mod, cos, sin, T, None, None, fwd_only=True)
        x_tokens[b * S:(b + 1) * S].copy_(X)

    # ------------------------------------------------------------------ whole-trunk backward of one sample
    def backward_sample(self, ckpt: torch.Tensor, b: int, mod_all: torch.Tensor, x_final_img: torch.Tensor,
                        dxn_img: torch.Tensor, T: int, N: int, hp: int, wp: int, grads: torch.Tensor,
                        dmod_out: Optional[torch.Tensor] = None, on_block_done=None) -> None:
        """ckpt [nblocks, B*S, D] block inputs of the last student forward; mod_all [B, n_mod]; x_final_img [N, D] the
        image tokens entering norm_out; dxn_img [N, D] the gradient at the velocity head's input.  Accumulates the
        LoRA gradients of sample b into ``grads``.  on_block_done(block): called after each block's backward (the distiller
        starts the all-reduce of that block's gradient slice when this is the iteration's last sample)."""
        D, S = self.D, T + N
        mod = mod_all[b]
        self.row0 = b * S
        self.dmod = dmod_out            # [n_mod] fp32 of this sample (zeroed by the caller), or None: no modulation gradients
        cos, sin = self.eng.rope_tables(hp, wp, T)
        fin = (self.nd * 12 + self.ns * 3) * D
        dX = torch.zeros(S, D, dtype=torch.bfloat16, device=self.dev)
        ops.ln_modulate_backward(x_final_img, dxn_img, mod[fin:fin + D], out=dX[T:])     # norm_out: scale first
        for i in reversed(range(self.ns)):
            X = ckpt[self.nd + i, b * S:(b + 1) * S]
            dX = self._single_block(i, X, mod, cos, sin, T, dX, grads)
            if on_block_done is not None:
                self._join_wgrad()
                on_block_done(self.nd + i)
        for i in reversed(range(self.nd)):
            X = ckpt[i, b * S:(b + 1) * S]
            dX = self._double_block(i, X, mod, cos, sin, T, dX, grads)
            if on_block_done is not None:
                self._join_wgrad()
                on_block_done(i)
        self._join_wgrad()
