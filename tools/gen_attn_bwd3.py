#!/usr/bin/env python3
"""Generator of the instruction stream of the key-stationary dK / dV kernel (arcflow_amd/csrc/afx_attn_bwd3.hip, round 5).

Same method as tools/gen_attn3.py (the forward): one wave per SIMD, every instruction of the loop its own `asm volatile` statement (source
order = issue order), every wide operand ASM-OWNED (literal register names; `amdgpu_num_vgpr(60)` confines hipcc to v[0:59], arcflow_amd/build.py
audits the ISA).  A wave owns 32 keys (lane = key); the queries stream by in HALVES of 32 (two per 64-query tile):

    accumulator file   a[  0: 63]  dV^T accumulators  DVA[d]      = a[16 d ..+15]          (d = 32-column block of the head dim)
                       a[ 64:127]  dK^T accumulators  DKA[d]      = a[64 + 16 d ..+15]
                       a[128:159]  K fragments        KF[s]       = a[128 + 4 s ..+3]      (stationary B operands, s = 16-wide k-step of d)
                       a[160:191]  V fragments        VF[s]       = a[160 + 4 s ..+3]
                       a[192:255]  row fragments      RQ[s] / RO[s] = a[192 + 4 s ..] / a[224 + 4 s ..]   (Q / dO rows of the half: A operands of S, dP)
    arch VGPRs         v[  0: 59]  hipcc's: LDS addresses, loop control          v[60:63]  per-lane source offsets of the four DMA pieces (B3_SETX)
                       v[ 64:127]  S / dP             SS[set] = v[64 + 32 set ..+15], DP[set] = v[80 + 32 set ..+15]   (set = half & 1)
                       v[128:191]  transposed fragments TO[i] = v[128 + 4 i ..+3] (dO^T), TQ[i] = v[160 + 4 i ..+3] (Q^T), i = 4 ksub + d
                       v[192:223]  P / dS words       PW[set][ksub] = v[192 + 16 set + 4 ksub ..+3], DW[set][ksub] = v[200 + 16 set + 4 ksub ..+3]
                       v[224:255]  L (16) | delta (16) of the half whose softmax gradient runs

One PHASE = 32 MFMAs = one half-step p of the software pipeline (J = p % 8 = the half's ring slot; set = p & 1):
    MFMA   SD(p):   S(p) = Q(p) K^T, dP(p) = dO(p) V^T          (16, A = row fragments read in phase p - 1)
           DV(p-2): dV^T += dO^T(p-2) P(p-2), dK^T += Q^T(p-2) dS(p-2)   (16, A = ds_read_b64_tr_b16 fragments read in this phase's first half)
    VALU   SM(p-1): P = exp2(S c - L), dS = P dP', both rounded to bf16 words (64 instructions, in place in the S / dP registers); dP' = dP - delta comes out of
                    the MFMA chain itself: the chain's C operand is the half's -delta, read from LDS straight into the accumulator's registers
    LDS    32 transpose reads of half p - 2 (gaps 0-15), 16 row reads of half p + 1, 4 L reads of half p, 4 -delta reads of half p + 1 (gaps 16-31)
    DMA    half p + 5 -> slot (J + 5) % 8: 4 pieces of the wave's tensor (waves 0, 1: Q rows 0-15 / 16-31, waves 2, 3: dO) + the 256-byte L | delta block
Every phase issues the SAME sequence of LDS reads and DMA pieces, so the counted waits (lgkmcnt / vmcnt) hold in the peeled first and last
phases too (which only drop MFMAs / VALU work whose inputs do not exist yet).

LDS half-slot (16640 bytes): Q [4 d-blocks][32 rows][64 B] | dO (same) | L[32] | delta[32].  Inside a 64-byte row the 16-byte chunk c sits at
c ^ ((row >> 2) & 3): a ds_read_b128 of 16 consecutive rows and a transpose read of 4 rows x 64 bytes both touch every bank once.

--mode dq: the QUERY-stationary dQ kernel is the same stream with the tensors' roles swapped (lane = query; K / V halves stream by):
    SD(p): S^T = K Q^T, dP^T = V dO^T (A = K / V rows, B = the wave's Q / dO rows, stationary in a[128:191])
    SM(p-1): dS^T = P^T dP'^T only (L and -delta are per LANE: v224, and v[144:159] = 16 copies of -delta as the dP^T chain's C operand)       DV(p-2): dQ^T += K^T dS^T (8 MFMAs, A = transpose reads of the K half)
24 MFMAs, 56 VALU, 32 LDS reads and 4 DMA pieces per phase; half-slot = K | V = 16384 bytes; files q3_*.inc.

Writes arcflow_amd/csrc/gen/b3_*.inc / q3_*.inc (committed; the build does not run this script).  Usage: python tools/gen_attn_bwd3.py [--mode dq]
"""
import argparse
import os

SLOT = 16640
MODE = 'dkv'            # 'dq': see the header
X_DO = 8192
STAT = 16384
NSLOT = 8
MFMA = 'v_mfma_f32_32x32x16_bf16'
ABL = set()


def rng(prefix, lo, n):
    return f'{prefix}[{lo}:{lo + n - 1}]'


def DVA(d):
    return rng('a', 16 * d, 16)


def DKA(d):
    return rng('a', 64 + 16 * d, 16)


def KF(s):
    return rng('a', 128 + 4 * s, 4)


def VF(s):
    return rng('a', 160 + 4 * s, 4)


def RQ(s):
    return rng('a', 192 + 4 * s, 4)


def RO(s):
    return rng('a', 224 + 4 * s, 4)


def SS(st):
    return rng('v', 64 + 32 * st, 16)


def DP(st):
    return rng('v', 80 + 32 * st, 16)


def SSx(st, r):
    return f'v{64 + 32 * st + r}'


def DPx(st, r):
    return f'v{80 + 32 * st + r}'


def T_lo(which, i):
    """first register of transposed fragment i = 4 ksub + d: which 0 = dO^T (TO), 1 = Q^T (TQ)"""
    return (128 if which == 0 else 160) + 4 * i


def PW(st, ks):
    return rng('v', 192 + 16 * st + 4 * ks, 4)


def DW(st, ks):
    return rng('v', 200 + 16 * st + 4 * ks, 4)


def PWx(st, w):
    return f'v{192 + 16 * st + w}'


def DWx(st, w):
    return f'v{200 + 16 * st + w}'


def Lx(r):
    return f'v{224 + r}'


def Dx(r):
    return f'v{240 + r}'


def asm(text, outs='', ins='', clob=''):
    s = f'asm volatile("{text}" : {outs} : {ins}'
    if clob:
        s += f' : {clob}'
    return s + ');'


def wait(text, own=False):
    return asm(text, '', '', '"memory"' + (', "v255", "a255"' if own else ''))


# ---- register bookkeeping of the LDS queue: which read (sequence number) last wrote a register ----------------------------------------
class Lds:
    """LDS reads return in order: a consumer of read q may go ahead once at most (issued - 1 - q) younger reads are outstanding.  lgkmcnt is a 4-bit
    counter: counts above 15 are clamped (an over-wait, by then long satisfied: such reads were issued >= 8 MFMAs earlier)."""

    def __init__(self):
        self.issued = 0
        self.writer = {}
        self.done = -1

    def read(self, regs):
        for r in regs:
            self.writer[r] = self.issued
        self.issued += 1

    def need(self, regs):
        q = max((self.writer.get(r, -1) for r in regs), default=-1)
        if q <= self.done:
            return []
        n_after = min(self.issued - 1 - q, 15)
        self.done = self.issued - 1 - n_after
        return [wait(f's_waitcnt lgkmcnt({n_after})')]

    def new_phase(self):
        self.done = -1           # nothing carried: every phase re-establishes what it needs (peeled phases emit fewer waits than the steady ones)


def regs_of(prefix, lo, n):
    return [f'{prefix}{i}' for i in range(lo, lo + n)]


def mfma_variant(stmt):
    """--ablate mfma16 (timing / power experiment, WRONG results): every 32x32x16 MFMA becomes two 16x16x32 MFMAs of the same operands (same flops, the smaller
    shape's energy per flop: a pure 16x16x32 loop runs 14 % faster under the power cap) writing the first eight registers of its accumulator."""
    if 'mfma16' not in ABL or MFMA not in stmt:
        return [stmt]
    import re
    m = re.search(r'v_mfma_f32_32x32x16_bf16 ([av])\[(\d+):\d+\], (\S+), (\S+), (\S+?)"', stmt)
    bank, lo, a, b, c = m.group(1), int(m.group(2)), m.group(3), m.group(4), m.group(5)
    out = []
    for h in range(2):
        d = f'{bank}[{lo + 4 * h}:{lo + 4 * h + 3}]'
        cc = c if c == '0' else d
        out.append(asm(f'v_mfma_f32_16x16x32_bf16 {d}, {a}, {b}, {cc}'))
    return out


# ---- single instructions ---------------------------------------------------------------------------------------------------------------
def sd(st, m):
    """MFMA m = 0..15 of S = Q K^T (even) / dP = dO V^T (odd), k-step s = m >> 1"""
    s = m >> 1
    if m & 1:
        # the dP chain starts from -delta instead of 0 (dP - delta for free): the dQ stream keeps the lane's -delta in a constant 16-register block,
        # the dK / dV stream reads the half's -delta straight into the accumulator's registers (read_negdelta)
        c0 = NEGD if MODE == 'dq' else DP(st)
        regs = regs_of('a', 224 + 4 * s, 4) + (regs_of('v', 80 + 32 * st, 16) if s == 0 and MODE != 'dq' else [])
        return asm(f'{MFMA} {DP(st)}, {RO(s)}, {VF(s)}, {c0 if s == 0 else DP(st)}'), regs
    return asm(f'{MFMA} {SS(st)}, {RQ(s)}, {KF(s)}, {0 if s == 0 else SS(st)}'), regs_of('a', 192 + 4 * s, 4)


def dv(st, n):
    """MFMA n = 0..15 of dV^T += dO^T P (even) / dK^T += Q^T dS (odd): ksub = n >> 3, d = (n >> 1) & 3 (an accumulator comes back after 8 MFMAs)"""
    if MODE == 'dq':         # n = 0..7: dQ^T[d] += K^T(ks, d) dS^T(ks)
        ks, d = n >> 2, n & 3
        lo = T_lo(1, 4 * ks + d)
        return asm(f'{MFMA} {DVA(d)}, {rng("v", lo, 4)}, {DW(st, ks)}, {DVA(d)}'), regs_of('v', lo, 4)
    ks, d, which = n >> 3, (n >> 1) & 3, n & 1
    lo = T_lo(which, 4 * ks + d)
    if which == 0:
        return asm(f'{MFMA} {DVA(d)}, {rng("v", lo, 4)}, {PW(st, ks)}, {DVA(d)}'), regs_of('v', lo, 4)
    return asm(f'{MFMA} {DKA(d)}, {rng("v", lo, 4)}, {DW(st, ks)}, {DKA(d)}'), regs_of('v', lo, 4)


NEGD = 'v[144:159]'       # dq mode: 16 copies of the lane's -delta (C operand of the first dP^T MFMA of every half)


def ring(slot):
    return 'h' if slot >= 4 else 'l'


def read_row(lds, m, slot, extra=None):
    """row fragment of next phase's SD MFMA m: tensor m & 1 (0 Q, 1 dO), k-step s = m >> 1: row ql, logical chunk 2 (s & 1) + hi of d-block s >> 1"""
    s, tensor = m >> 1, m & 1
    lo = (224 if tensor else 192) + 4 * s
    off = tensor * X_DO + (s >> 1) * 2048
    lds.read(regs_of('a', lo, 4))
    if extra:
        return asm(f'ds_read_b128 {rng("a", lo, 4)}, %0 offset:{off}', '', f'"v"(raddr{s & 1}l + {extra})')
    return asm(f'ds_read_b128 {rng("a", lo, 4)}, %0 offset:{(slot & 3) * SLOT + off}', '', f'"v"(raddr{s & 1}{ring(slot)})')


def read_tr(lds, k, slot, extra=None):
    """transpose read k = 0..31: fragment of DV MFMA n = k >> 1, half rd = k & 1 (rows 16 ksub + 8 rd + 4 hi + 0..3 of the half)"""
    n, rd = k >> 1, k & 1
    if MODE == 'dq':
        ks, d, which = n >> 2, n & 3, 1          # K^T fragments: the K half sits first in the slot (as Q in the dK / dV kernel)
    else:
        ks, d, which = n >> 3, (n >> 1) & 3, n & 1
    lo = T_lo(which, 4 * ks + d) + 2 * rd
    off = (X_DO if which == 0 else 0) + d * 2048 + 16 * ks * 64          # (the + 8 rows of rd = 1 sit in taddr1*: the 16-bit offset field)
    lds.read(regs_of('v', lo, 2))
    if extra:
        return asm(f'ds_read_b64_tr_b16 {rng("v", lo, 2)}, %0 offset:{off}', '', f'"v"(taddr{rd}l + {extra})')
    return asm(f'ds_read_b64_tr_b16 {rng("v", lo, 2)}, %0 offset:{(slot & 3) * SLOT + off}', '', f'"v"(taddr{rd}{ring(slot)})')


def read_stat(lds, which, g, slot):
    """L (which 0) / delta (1) of queries 8 g + 4 hi + 0..3 of the half -> registers 4 g ..+3 of the block"""
    lo = (224 if which == 0 else 240) + 4 * g
    lds.read(regs_of('v', lo, 4))
    return asm(f'ds_read_b128 {rng("v", lo, 4)}, %0 offset:{(slot & 3) * SLOT + which * 128 + 32 * g}', '', f'"v"(saddr{ring(slot)})')


def read_negdelta(lds, g, slot, st):
    """-delta of queries 8 g + 4 hi + 0..3 of the half in `slot` -> registers 4 g ..+3 of the dP accumulator of set st (its MFMA chain starts from them)"""
    lo = 80 + 32 * st + 4 * g
    lds.read(regs_of('v', lo, 4))
    return asm(f'ds_read_b128 {rng("v", lo, 4)}, %0 offset:{(slot & 3) * SLOT + 128 + 32 * g}', '', f'"v"(saddr{ring(slot)})')


def dma(piece, slot):
    # (s_add_u32 writes SCC: without the clobber hipcc keeps a loop-exit compare alive across the statement)
    if piece < 4 or MODE == 'dq':
        # (the per-lane source offsets of the four pieces are asm-owned too: v60..v63, set by B3_SETX -- as C++ values hipcc copied the set every phase)
        return asm(f's_add_u32 m0, %0, {slot * SLOT + piece * 2048}\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 v{60 + piece}, %1', '',
                   '"s"(wave_lds), "s"(xsrc)', '"memory", "scc"')
    if STAT1:
        # --stat-one-wave: the 256-byte L | -delta block is one DMA instruction: wave (slot & 3) alone issues it (EXEC = 0 in the other three: the instruction is a no-op there)
        return asm(f's_add_u32 m0, %0, {slot * SLOT + STAT}\\n\\ts_cmp_eq_u32 %3, {slot & 3}\\n\\ts_cselect_b64 exec, -1, 0\\n\\tglobal_load_lds_dword %1, %2\\n\\ts_mov_b64 exec, -1', '',
                   '"s"(lds0), "v"(sofs), "s"(ssrc), "s"(wave)', '"memory", "scc"')
    return asm(f's_add_u32 m0, %0, {slot * SLOT + STAT}\\n\\ts_nop 0\\n\\tglobal_load_lds_dword %1, %2', '', '"s"(lds0), "v"(sofs), "s"(ssrc)',
               '"memory", "scc"')


# ---- the softmax gradient of one half as an instruction list ----------------------------------------------------------------------------
def sm_ops(st):
    """(text, registers read that come from LDS, stage).  Stage k: A_k  S = S c - L;  B_(k-1)  S = exp2(S);  D_(k-3)  dP = S dP  (dP arrives as dP - delta: see sd());
    E  words of pairs (k - 4, k - 3).  A v_exp_f32 result is never read by the next instruction (trans -> VALU wait state)."""
    ops = []
    dq = MODE == 'dq'
    for k in range(16 + 3):
        if k < 16:
            ops.append((f'v_fma_f32 {SSx(st, k)}, {SSx(st, k)}, %0, -{Lx(0 if dq else k)}', [] if dq else [Lx(k)], k, True))
        if 0 <= k - 1 < 16:
            ops.append((f'v_exp_f32 {SSx(st, k - 1)}, {SSx(st, k - 1)}', [], k, False))
        if 0 <= k - 3 < 16:
            i = k - 3
            ops.append((f'v_mul_f32 {DPx(st, i)}, {SSx(st, i)}, {DPx(st, i)}', [], k, False))
            if i & 1:
                if not dq:
                    ops.append((f'v_cvt_pk_bf16_f32 {PWx(st, i >> 1)}, {SSx(st, i - 1)}, {SSx(st, i)}', [], k, False))
                ops.append((f'v_cvt_pk_bf16_f32 {DWx(st, i >> 1)}, {DPx(st, i - 1)}, {DPx(st, i)}', [], k, False))
    assert len(ops) == (56 if dq else 64)
    if 'novalu' in ABL:
        ops = []
    return ops


def emit_valu(lds, op):
    text, lregs, _, uses_c = op
    out = lds.need(lregs) if lregs else []
    out.append(asm(text, '', '"s"(c)' if uses_c else ''))
    return out


# ---- one phase -------------------------------------------------------------------------------------------------------------------------------
VALU_PER_GAP = 3
DMA_GAPS = None           # --dma-gaps: the MFMA gaps that carry the phase's DMA pieces
STAT1 = False             # --stat-one-wave: see dma()
BAR2 = False              # --barrier-every 2: wait + barrier at even phases only; a phase then fetches half p + 4 (see phase())


def lead():
    return 4 if BAR2 else 5


def slot_bytes():
    return 16384 if MODE == 'dq' else SLOT


def pieces():
    return 4 if MODE == 'dq' else 5


def vm_pieces():
    """pieces per phase the counted vmcnt waits may assume outstanding: with --stat-one-wave a wave issues 4 or 5 (or, if the hardware does not count an EXEC = 0
    instruction, 4): waiting down to 4 per phase is safe in every case"""
    return 4 if STAT1 else pieces()


def phase(lds, J, do_sd=True, do_sm=True, do_dv=True, trace=None):
    """half-step p with ring slot J = p % 8 (set = J & 1).  The C++ around it provides xsrc / ssrc (DMA of half p + lead(); the per-lane offsets are v60..v63)."""
    st = J & 1
    dq = MODE == 'dq'
    ngap = 24 if dq else 32
    lds.new_phase()
    out = [f'// ---- phase J = {J}: SD(p){"" if do_sd else " [off]"} | SM(p-1){"" if do_sm else " [off]"} | DV(p-2){"" if do_dv else " [off]"}']
    # everything but the last three phases' DMA pieces has landed (half p + 1 was issued in phase p - 4); the barrier makes it everybody's pieces and
    # says every wave is done with phase p - 1 (slot (J + 5) % 8 = half p - 3: last read there)
    if trace is not None:
        out.append(f'B3_TR({trace}, 0)')
    if not BAR2:
        out.append(wait(f's_waitcnt vmcnt({3 * vm_pieces()})', own=True))
        out.append(wait('s_barrier'))
    elif J % 2 == 0:
        # --barrier-every 2: at an even phase p only the pieces of phase p - 1 (half p + 3) may be in flight: halves <= p + 2 have landed, which is what phases p and
        # p + 1 read; phase u fetches half u + 4 into the slot of half u - 4, whose last (transpose) reads were issued in phase u - 2 -- before this barrier or the last one
        out.append(wait(f's_waitcnt vmcnt({vm_pieces()})', own=True))
        out.append(wait('s_barrier'))
    else:
        out.append(wait('s_nop 0', own=True))
    if trace is not None:
        out.append(f'B3_TR({trace}, 1)')
    if dq and do_sm:
        # ragged key range: the scores of the keys past S (clamped copies of the last K row) -> -inf before the half's softmax gradient reads them (cold)
        out.append(f'if (__builtin_expect(ragged && p - 1 >= NH - 2, 0)) {{ const int kb_ = 32 * (p - 1); B3_MASK_S{1 - st} }}')
    sm = sm_ops(1 - st) if do_sm else []
    mem = {g: [] for g in range(ngap)}                                       # gap -> memory instructions (callables: they register with the queue model in issue order)
    if dq:
        for g in range(16):                                                 # 16 transpose reads of K(p - 2); row read m of half p + 1 two gaps behind the MFMA that read the old fragment
            mem[g].append(lambda g=g: read_tr(lds, g, (J + 6) % NSLOT))
        for m in range(16):
            mem[m + 2].append(lambda m=m: read_row(lds, m, (J + 1) % NSLOT))
        dma_gaps = dict(zip(DMA_GAPS or (17, 19, 21, 23), range(4)))
    else:
        # gap in which the last instruction of stage 4 g + 3 (the last reader of L group g) is issued; the softmax gradient's last instruction
        pos, stage_gap, last_gap = 0, {}, -1
        for gap in range(ngap):
            for op in sm[pos:pos + VALU_PER_GAP]:
                stage_gap[op[2]] = gap
                last_gap = gap
            pos += VALU_PER_GAP
        free_after = [stage_gap.get(4 * g + 3, -1) for g in range(4)]      # group g's registers may be overwritten from the NEXT gap on
        for g in range(16):
            mem[g].append(lambda g=g: read_tr(lds, 2 * g, (J + 6) % NSLOT))
            mem[g].append(lambda g=g: read_tr(lds, 2 * g + 1, (J + 6) % NSLOT))
            mem[16 + g].append(lambda g=g: read_row(lds, g, (J + 1) % NSLOT))
        gap = 16
        for g in range(4):                                                  # L of half p -> v[224:239] (SM(p) runs in the next phase)
            gap = max(gap, free_after[g] + 1)
            assert gap < ngap, 'no room for the L reads'
            mem[gap].append(lambda g=g: read_stat(lds, 0, g, J))
            gap += 1
        # -delta of half p + 1 -> the dP accumulator of set 1 - st, once SM(p - 1) is done with it (the first dP MFMA of the next phase starts from it)
        gap = max(ngap - 4, last_gap + 1)
        assert gap + 3 < ngap, 'no room for the -delta reads'
        for g in range(4):
            mem[gap + g].append(lambda g=g: read_negdelta(lds, g, (J + 1) % NSLOT, 1 - st))
        dma_gaps = dict(zip(DMA_GAPS or (17, 20, 23, 26, 29), range(5)))
    k = 0
    for gap in range(ngap):
        if gap < 16:
            if do_sd:
                text, regs = sd(st, gap)
                out += lds.need(regs)
                out += mfma_variant(text)
        elif do_dv:
            text, regs = dv(st, gap - 16)
            out += lds.need(regs)
            out += mfma_variant(text)
        if 'nolds' not in ABL:
            ops_ = [f() for f in mem[gap]]                       # (every read registers with the queue model)
            if 'halfmem' in ABL:                                 # timing experiment (WRONG results): what a 64-row wave -- every fragment read feeding two MFMAs, half the
                ops_ = ops_[::2] if len(ops_) > 1 else (ops_ if gap % 2 == 0 else [])      # DMA pieces per MFMA -- would leave of the memory instructions
            out += ops_
        if gap in dma_gaps and 'nodma' not in ABL and not ('halfmem' in ABL and dma_gaps[gap] in (1, 3)):
            out.append(dma(dma_gaps[gap], (J + lead()) % NSLOT))
        for op in sm[k:k + VALU_PER_GAP]:
            out += emit_valu(lds, op)
        k += VALU_PER_GAP
        if trace is not None and gap == 15:
            out.append(f'B3_TR({trace}, 2)')
    assert k >= len(sm)
    return out


# ---- blocks ------------------------------------------------------------------------------------------------------------------------------------
HDR = '// generated by tools/gen_attn_bwd3.py'


def init():
    out = [f'{HDR} -- accumulators = 0; B3_LEAD: a phase fetches the half that many phases ahead', '#undef B3_LEAD', f'#define B3_LEAD {lead()}']
    for lo in range(0, 64 if MODE == 'dq' else 128, 16):
        out.append(asm('\\n\\t'.join(f'v_accvgpr_write_b32 a{lo + r}, 0' for r in range(16)), '', '', '"v255", "a255"' if lo == 0 else ''))
    return out


def kv_loads():
    out = [f'{HDR} -- the wave\'s stationary rows (K / V; dq mode: Q / dO) straight into the accumulator file (B operands); waited for by hand']
    for s in range(8):
        out.append(asm(f'global_load_dwordx4 {KF(s)}, %0, off offset:{s * 32}', '', '"v"(kptr)', '"memory"'))
        out.append(asm(f'global_load_dwordx4 {VF(s)}, %0, off offset:{s * 32}', '', '"v"(vptr)', '"memory"'))
    if MODE == 'dq':         # L and delta of the lane's query (padded side array: +inf | 0 past S)
        out.append(asm(f'global_load_dword {Lx(0)}, %0, off', '', '"v"(lptr)', '"memory"'))
        out.append(asm(f'global_load_dword {Dx(0)}, %0, off offset:128', '', '"v"(lptr)', '"memory"'))
    return out


def prologue_dma():
    out = [f'{HDR} -- halves 0..{lead() - 1} -> slots 0..{lead() - 1} ({pieces()} pieces each: the order the counted waits assume)']
    for u in range(lead()):
        out.append(f'{{ B3_SRC({u}) B3_SETX(xofs0, xofs1, xofs2, xofs3)')
        out += [dma(i, u) for i in range(pieces())]
        out.append('}')
    return out


def first_rows(lds):
    out = [f'{HDR} -- the stationary rows\' loads + 5 halves of DMA pieces in flight: everything up to half 0 has landed when only halves 1..4 remain']
    out.append(wait(f's_waitcnt vmcnt({(lead() - 1) * vm_pieces()})', own=True))
    out.append(wait('s_barrier'))
    out += [read_row(lds, m, 0) for m in range(16)]
    if MODE == 'dq':
        out.append(asm('\\n\\t'.join(f'v_mov_b32 v{144 + r}, {Dx(0)}' for r in range(16))))       # the constant -delta block (stats hold -delta)
    else:
        out += [read_negdelta(lds, g, 0, 0) for g in range(4)]
    out.append(wait('s_waitcnt lgkmcnt(0)'))
    return out


def tail(lds, which):
    """the two draining phases (runtime ring slot through `ts`, a byte offset added to the ring-0 addresses): not pipelined.
    which 0: p = NH:     SM(NH - 1), DV(NH - 2)       which 1: p = NH + 1: DV(NH - 1)"""
    out = [f'{HDR} -- draining phase {which} (set = {which}: NH is even)']
    st = which                                   # p = NH + which, set = p & 1
    out += [read_tr(lds, k, 0, 'ts') for k in range(16 if MODE == 'dq' else 32)]
    if which == 0:
        lds.new_phase()
        if MODE == 'dq':
            out.append(f'if (ragged) {{ const int kb_ = 32 * (NH - 1); B3_MASK_S{1 - st} }}')
        for op in sm_ops(1 - st):
            out += emit_valu(lds, op)
    out.append(wait('s_waitcnt lgkmcnt(0)\\n\\ts_nop 3'))
    out += [dv(st, n)[0] for n in range(8 if MODE == 'dq' else 16)]
    return out


def mask_ops(st):
    """dq mode, ragged key range (cold): S^T register r of a lane holds key 32 half + (r & 3) + 8 (r >> 2) + 4 hi (the MFMA D layout); kb_ = 32 half."""
    lines = [f'#define B3_MASK_S{st} \\']
    for r in range(16):
        key = (r & 3) + 8 * (r >> 2)
        lines.append(f'  {{ const float pen_ = kb_ + {key} + 4 * hi >= S ? -INFINITY : 0.f; asm volatile("v_add_f32 {SSx(st, r)}, {SSx(st, r)}, %0" : : "v"(pen_)); }} \\')
    lines.append('  asm volatile("s_nop 1");')
    return lines


def readout():
    out = [f'{HDR} -- epilogue: one [32 d][32 lanes] accumulator tile -> 16 VGPR scalars ox[0..15]']
    pre = 'Q3' if MODE == 'dq' else 'B3'
    out.append(f'#define {pre}_DRAIN ' + asm('s_waitcnt vmcnt(0)\\n\\ts_nop 15\\n\\ts_nop 15', '', '', '"memory"'))
    for name, base in ((('Q', 0),) if MODE == 'dq' else (('V', 0), ('K', 64))):
        for d in range(4):
            lo = base + 16 * d
            text = '\\n\\t'.join(f'v_accvgpr_read_b32 %{r}, a{lo + r}' for r in range(16))
            outs = ', '.join(f'"=v"(ox[{r}])' for r in range(16))
            out.append(f'#define {pre}_READ_{name}_{d} ' + asm(text, outs, ''))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='dkv', choices=['dkv', 'dq'])
    ap.add_argument('--out', default='gen', help='directory under arcflow_amd/csrc (afx_attn_bwd3.hip includes B3_GEN/..., default gen)')
    ap.add_argument('--ablate', default='', help='comma list of nodma, nolds, novalu, mfma16, halfmem: timing experiments, WRONG results')
    ap.add_argument('--valu-per-gap', type=int, default=3)
    ap.add_argument('--barrier-every', type=int, default=1, choices=[1, 2])
    ap.add_argument('--stat-one-wave', action='store_true')
    ap.add_argument('--dma-gaps', default='', help='comma list: the MFMA gaps of a phase that carry its DMA pieces (5 for dkv, 4 for dq)')
    a = ap.parse_args()
    ABL.update(x for x in a.ablate.split(',') if x)
    global VALU_PER_GAP, MODE, SLOT, DMA_GAPS, BAR2, STAT1
    STAT1 = a.stat_one_wave
    BAR2 = a.barrier_every == 2
    DMA_GAPS = tuple(int(x) for x in a.dma_gaps.split(',')) if a.dma_gaps else None
    VALU_PER_GAP = a.valu_per_gap
    MODE = a.mode
    SLOT = slot_bytes()
    px = 'q3' if MODE == 'dq' else 'b3'
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'arcflow_amd', 'csrc', a.out)
    os.makedirs(root, exist_ok=True)
    files = {f'{px}_init.inc': init(), f'{px}_kvload.inc': kv_loads(), f'{px}_prologue_dma.inc': prologue_dma(), f'{px}_readout.inc': readout()}
    if MODE == 'dq':
        files[f'{px}_mask.inc'] = mask_ops(0) + mask_ops(1)
    lds = Lds()
    files[f'{px}_first_rows.inc'] = first_rows(lds)
    files[f'{px}_p0.inc'] = [f'{HDR} -- p = 0 (slot 0): SD(0) only'] + phase(lds, 0, True, False, False)
    files[f'{px}_p1.inc'] = [f'{HDR} -- p = 1 (slot 1): SD(1), SM(0)'] + phase(lds, 1, True, True, False)
    # steady state: the loop enters at J = 2 and walks 2, 3, ..., 7, 0, 1; every phase issues the same LDS sequence, so one running queue model serves
    for J in (2, 3, 4, 5, 6, 7, 0, 1):
        files[f'{px}_body{J}.inc'] = [f'{HDR} -- steady-state phase, ring slot J = {J}'] + phase(lds, J, trace=J)
    files[f'{px}_tail0.inc'] = tail(lds, 0)
    files[f'{px}_tail1.inc'] = tail(lds, 1)
    for f in os.listdir(root):
        if f.startswith(px + '_') and f not in files:
            os.remove(os.path.join(root, f))
    for name, lines in files.items():
        with open(os.path.join(root, name), 'w') as f:
            f.write('\n'.join(lines) + '\n')
        print(name, len(lines), 'lines')


if __name__ == '__main__':
    main()
