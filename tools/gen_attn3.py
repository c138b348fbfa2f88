#!/usr/bin/env python3
"""Generator of the instruction stream of the one-wave-per-SIMD attention kernel (arcflow_amd/csrc/afx_attn3.hip).

The kernel's main loop is a hand-placed stream: every instruction is its own `asm volatile` statement, so the
source order is the issue order.  All wide operands are ASM-OWNED: they appear by literal register name in the
instruction text, hipcc never sees a variable behind them.  The kernel carries `amdgpu_num_vgpr(96)`, which
confines hipcc's own allocation to v[0:95] (and a[0:95], which it must never need: arcflow_amd/build.py audits the
ISA for accumulator moves / scratch traffic outside the asm statements); everything above is this file's:

    accumulator file   a[  0:127]  O^T accumulators   O[slab][d]   = a[64 slab + 16 d ..+15]
                       a[128:191]  Q^T fragments      Q[slab][s]   = a[128 + 32 slab + 4 s ..+3]
                       a[192:255]  K fragments        K[kb][s]     = a[192 + 32 kb + 4 s ..+3]
    arch VGPRs         v[  0: 95]  hipcc's: addresses, running max / row sums, the softmax stream's temporaries
                       v[ 96:159]  S^T accumulators   S[slab][kb]  = v[96 + 32 slab + 16 kb ..+15]
                       v[160:223]  V^T fragments      VF[i]        = v[160 + 4 i ..+3], i = 4 g + d (order of use)
                       v[224:255]  P^T words          W[slab][i]   = v[224 + 16 slab + i]

What was tried first, and what hipcc did with it (ROCm 7.2), for the record:
  * "a" / "v" constraints on C++ variables, allocation left to hipcc: 464 spilled registers, 2600 accumulator moves in the loop
    (461 of 512 registers are spoken for; the greedy allocator splits live ranges of 16-register tuples when it is this full);
  * explicit-register constraints ("{a[0:15]}") on C++ variables: 728 spilled -- between two statements a pinned value is an
    ordinary virtual register again, long-lived scalars get parked inside pinned ranges and are spilled when the range comes alive;
  * accumulator file by literal names only, arch side by generic constraints: correct allocation but hipcc parks live-through
    arch values in accumulator registers it believes free (silent corruption of Q fragments), and with clobbers telling it not
    to, V^T fragments are spilled to scratch inside the loop (a scratch reload is a VMEM op with its own s_waitcnt vmcnt(0)).

It writes arcflow_amd/csrc/gen/a3_*.inc (committed; the build does not run this script):
    a3_init.inc      O^T = 0
    a3_qload.inc, a3_prologue_dma.inc   the prologue's loads
    a3_tile0.inc     tile 0, not pipelined
    a3_body{0..3}.inc one loop iteration with ring slot J (see the phase table in afx_attn3.hip)
    a3_final.inc     the last tile
    a3_rescale.inc   cold path: O^T of a slab times alpha (two macros)
    a3_readout.inc   accumulator file <-> VGPR scalars, one 32-row tile of O^T at a time (macros for the epilogue / the hand-over init)
Usage: python tools/gen_attn3.py
"""
import argparse
import os

# ablation switches (timing experiments only -- results are wrong): see main()
ABL = set()
MERGE = False       # --merge: all VALU instructions of an MFMA gap in one asm statement

TILE = 16384
V_BASE = 4 * TILE
MFMA = 'v_mfma_f32_32x32x16_bf16'


def rng(prefix, lo, n):
    return f'{prefix}[{lo}:{lo + n - 1}]'


def O(sl, d):
    return rng('a', 64 * sl + 16 * d, 16)


def Q(sl, s):
    return rng('a', 128 + 32 * sl + 4 * s, 4)


def K(kb, s):
    return rng('a', 192 + 32 * kb + 4 * s, 4)


def S(sl, kb):
    return rng('v', 96 + 32 * sl + 16 * kb, 16)


def Sx(sl, i):
    """score i = 16 kb + r of slab sl as a scalar register"""
    return f'v{96 + 32 * sl + i}'


def VF(i):
    return rng('v', 160 + 4 * i, 4)


def W(sl, i):
    return f'v{224 + 16 * sl + i}'


def P(sl, g):
    return rng('v', 224 + 16 * sl + 4 * g, 4)


def asm(text, outs='', ins='', clob=''):
    s = f'asm volatile("{text}" : {outs} : {ins}'
    if clob:
        s += f' : {clob}'
    return s + ');'


# ---- single instructions -------------------------------------------------------------------------------------------------
def qk(sl, m):
    """MFMA m = 0..15 of S^T = K Q^T of slab sl: kb = m & 1, s = m >> 1."""
    kb, s = m & 1, m >> 1
    return asm(f'{MFMA} {S(sl, kb)}, {K(kb, s)}, {Q(sl, s)}, {0 if s == 0 else S(sl, kb)}')


def mask_ops(sl):
    """Ragged key range (KS keys in the segment, KS_pad = its tiles x 64), last tile only (cold): scores of the keys past the end -> -inf
    before the slab's softmax reads them.  Score register
    i = 16 kb + r of a lane holds key 32 kb + (r & 3) + 8 (r >> 2) + 4 hi of the tile (the MFMA D layout).  The K rows behind them are
    clamped copies of row S - 1 (finite), V^T is zero there."""
    lines = [f'#define A3_MASK_S{sl} \\']
    for i in range(32):
        kb, r = i >> 4, i & 15
        key = 32 * kb + (r & 3) + 8 * (r >> 2)
        lines.append(f'  {{ const float pen_ = last0 + {key} + 4 * hi >= KS ? -INFINITY : 0.f; '
                     f'asm volatile("v_add_f32 {Sx(sl, i)}, {Sx(sl, i)}, %0" : : "v"(pen_)); }} \\')
    lines.append('  asm volatile("s_nop 1");')
    return lines


def pv(sl, m):
    """MFMA m = 0..15 of O^T += V^T P^T of slab sl: d = m & 3, g = m >> 2."""
    d, g = m & 3, m >> 2
    return asm(f'{MFMA} {O(sl, d)}, {VF(4 * g + d)}, {P(sl, g)}, {O(sl, d)}')


def read_v(i, slot, extra=''):
    """V^T fragment i = 4 g + d of ring slot `slot` (immediate) -- or of a runtime slot through `extra` (an added VGPR)."""
    d, g = i & 3, i >> 2
    if extra:
        return asm(f'ds_read_b128 {VF(i)}, %0 offset:{d * 4096}', '', f'"v"(vaddr{g} + {extra})')
    return asm(f'ds_read_b128 {VF(i)}, %0 offset:{slot * TILE + d * 4096}', '', f'"v"(vaddr{g})')


def read_k(i, slot):
    """K fragment i = 2 s + kb of ring slot `slot`."""
    kb, s = i & 1, i >> 1
    return asm(f'ds_read_b128 {K(kb, s)}, %0 offset:{slot * TILE + kb * 8192}', '', f'"v"(kaddr{s})')


def dma(kind, slot, piece):
    base = (0 if kind == 'k' else V_BASE) + slot * TILE + piece * 4096
    src, off = ('ksrc', f'kofs{piece}') if kind == 'k' else ('vsrc', f'voff{piece}')
    # s_add_u32 writes SCC: without the clobber hipcc keeps a loop-exit compare alive across this statement (it did: endless loop)
    return asm(f's_add_u32 m0, %0, {base}\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 %1, %2', '', f'"s"(wave_lds), "v"({off}), "s"({src})',
               '"memory", "scc"')


# ---- the softmax of one slab and one tile as an instruction list --------------------------------------------------------------
class Op:
    """one VALU instruction: text with {0}, {1}, ... for its operands = [(variable, 'w' | 'r' | 'rw', 'v' | 's')], or raw C++"""

    def __init__(self, text, operands=(), raw=None):
        self.text, self.operands, self.raw = text, list(operands), raw


def emit(ops):
    """C++ for a run of Ops.  Consecutive instructions go into ONE asm statement: hipcc pads every boundary between two asm
    statements whose registers overlap with an s_nop (its hazard recogniser cannot look inside), ~45 issue slots per tile."""
    out, group = [], []

    def flush():
        if not group:
            return
        names, mode, cls = [], {}, {}
        for op in group:
            for v, m, c in op.operands:
                if v not in mode:
                    names.append(v)
                    cls[v] = c
                    mode[v] = {'w': 'ew', 'r': 'r', 'rw': 'rw'}[m]            # ew: written before any read -> early clobber
                elif m != 'r' and mode[v] == 'r':
                    mode[v] = 'rw'
        outs = [v for v in names if mode[v] != 'r']
        ins = [v for v in names if mode[v] == 'r']
        idx = {v: i for i, v in enumerate(outs + ins)}
        lines = [op.text.format(*[f'%{idx[v]}' for v, _, _ in op.operands]) for op in group]
        o = ', '.join(f'"{"=&" if mode[v] == "ew" else "+"}{cls[v]}"({v})' for v in outs)
        i = ', '.join(f'"{cls[v]}"({v})' for v in ins)
        out.append(asm('\\n\\t'.join(lines), o, i))
        group.clear()

    for op in ops:
        if op.raw is not None:
            flush()
            out.append(op.raw)
        else:
            group.append(op)
            if not MERGE:
                flush()
    flush()
    return out


def softmax_ops(sl):
    """The stream as Ops, one instruction each (the decision is plain C++ with a cold branch).  State: m / l0 / l1 of the slab."""
    X = 'AB'[sl]
    t0, t1, mb, nmc = f't{X}0', f't{X}1', f'mb{X}', f'nmc{X}'
    ops = []
    ops.append(Op(f'v_max3_f32 {{0}}, {Sx(sl, 0)}, {Sx(sl, 1)}, {Sx(sl, 2)}', [(t0, 'w', 'v')]))
    ops.append(Op(f'v_max3_f32 {{0}}, {Sx(sl, 16)}, {Sx(sl, 17)}, {Sx(sl, 18)}', [(t1, 'w', 'v')]))
    for j in range(6):
        ops.append(Op(f'v_max3_f32 {{0}}, {{0}}, {Sx(sl, 3 + 2 * j)}, {Sx(sl, 4 + 2 * j)}', [(t0, 'rw', 'v')]))
        ops.append(Op(f'v_max3_f32 {{0}}, {{0}}, {Sx(sl, 19 + 2 * j)}, {Sx(sl, 20 + 2 * j)}', [(t1, 'rw', 'v')]))
    ops.append(Op(f'v_max_f32 {{0}}, {{0}}, {Sx(sl, 15)}', [(t0, 'rw', 'v')]))
    ops.append(Op(f'v_max_f32 {{0}}, {{0}}, {Sx(sl, 31)}', [(t1, 'rw', 'v')]))
    ops.append(Op('v_max_f32 {0}, {0}, {1}', [(t0, 'rw', 'v'), (t1, 'r', 'v')]))
    ops.append(Op('v_mov_b32 {0}, {1}', [(mb, 'w', 'v'), (t0, 'r', 'v')]))
    # t0.hi <-> mb.lo: t0 = [lo, lo], mb = [hi, hi]   (VALU write -> permlane read: 2 wait states)
    ops.append(Op('s_nop 1\\n\\tv_permlane32_swap_b32 {0}, {1}', [(t0, 'rw', 'v'), (mb, 'rw', 'v')]))
    ops.append(Op('v_max_f32 {0}, {0}, {1}', [(t0, 'rw', 'v'), (mb, 'r', 'v')]))
    ops.append(Op('', raw=f'{{ const float m_new = fmaxf(m{X}, {t0}); '
                          f'if (__builtin_expect(__builtin_amdgcn_ballot_w64(m_new - m{X} > thr) != 0, 0)) {{ A3_RESCALE_{X}(m_new) }} }}'))
    ops.append(Op('v_mul_f32 {0}, {1}, {2}', [(nmc, 'w', 'v'), ('neg_c', 'r', 's'), (f'm{X}', 'r', 'v')]))
    for k in range(32 + 5):
        if k < 32:
            ops.append(Op(f'v_fma_f32 {{0}}, {Sx(sl, k)}, {{1}}, {{2}}', [(f'e{X}{k}', 'w', 'v'), ('c', 'r', 's'), (nmc, 'r', 'v')]))
        if 0 <= k - 2 < 32:
            i = k - 2
            ops.append(Op('v_exp_f32 {0}, {1}', [(f'p{X}{i}', 'w', 'v'), (f'e{X}{i}', 'r', 'v')]))
        if 0 <= k - 4 < 32:
            i = k - 4
            ops.append(Op('v_add_f32 {0}, {0}, {1}', [(f'l{X}{i & 1}', 'rw', 'v'), (f'p{X}{i}', 'r', 'v')]))
            if i & 1:
                ops.append(Op(f'v_cvt_pk_bf16_f32 {W(sl, i >> 1)}, {{0}}, {{1}}', [(f'p{X}{i - 1}', 'r', 'v'), (f'p{X}{i}', 'r', 'v')]))
    assert len(ops) == 134
    if 'noadd' in ABL:
        ops = [o for o in ops if 'v_add_f32' not in o.text]
    if 'noexp' in ABL:
        for o in ops:
            o.text = o.text.replace('v_exp_f32', 'v_mov_b32')
    if 'novalu' in ABL:
        ops = [o for o in ops if o.raw is not None or any(x in o.text for x in ('v_max', 'permlane', 'v_mov', 'v_mul'))]
    return ops


def valu_counts(mem_first_half):
    """VALU ops after MFMA m of a phase: 4 where the gap also carries a ds_read, 4 / 5 alternating elsewhere (sum >= 134)."""
    n = []
    for g in range(32):
        mem = g < 16 if mem_first_half else g >= 16
        n.append(4 if mem else (5 if g & 1 else 4))
    return n


def wait(text, own=False):
    """a wait / barrier statement; own: it names v255 / a255 as clobbered, which makes the kernel descriptor allocate the whole file"""
    return asm(text, '', '', '"memory"' + (', "v255", "a255"' if own else ''))


# ---- blocks ---------------------------------------------------------------------------------------------------------------------
def decl():
    out = ['// generated by tools/gen_attn3.py -- temporaries of the softmax stream (hipcc allocates them in v[0:95]); O^T = 0']
    for X in 'AB':
        out.append(f'float t{X}0, t{X}1, mb{X}, nmc{X};')
        out.append('float ' + ', '.join(f'e{X}{i}' for i in range(32)) + ';')
        out.append('float ' + ', '.join(f'p{X}{i}' for i in range(32)) + ';')
    for lo in range(0, 128, 16):
        out.append(asm('\\n\\t'.join(f'v_accvgpr_write_b32 a{lo + r}, 0' for r in range(16)), '', '', '"v255", "a255"' if lo == 0 else ''))
    return out


def q_loads():
    out = ['// generated by tools/gen_attn3.py -- Q^T fragments straight into the accumulator file; waited for by hand (a3_tile0.inc)']
    for s in range(8):
        for sl in range(2):
            out.append(asm(f'global_load_dwordx4 {Q(sl, s)}, %0, off offset:{s * 32}', '', f'"v"(qptr{sl})', '"memory"'))
    return out


def prologue_dma():
    out = ['// generated by tools/gen_attn3.py -- K(0), K(1), K(2), V^T(0), K(3), V^T(1): the order the counted waits assume']
    for kind, tile, slot in (('k', 0, 0), ('k', 1, 1), ('k', 2, 2), ('v', 0, 0), ('k', 3, 3), ('v', 1, 1)):
        src = 'ksrc' if kind == 'k' else 'vsrc'
        out.append(f'{{ const uint64_t {src} = {kind}_src({tile});' + (f' A3_KOFS({tile})' if kind == 'k' else ''))
        out += [dma(kind, slot, i) for i in range(4)]
        out.append('}')
    return out


def tile0():
    out = ['// generated by tools/gen_attn3.py -- tile 0, not pipelined']
    out.append('// 16 Q loads + 24 DMA pieces in flight: Q and K(0) have landed when <= 20 remain')
    out.append(wait('s_waitcnt vmcnt(20)\\n\\ts_barrier', own=True))
    out += [read_k(i, 0) for i in range(16)]
    out.append(wait('s_waitcnt lgkmcnt(0)'))
    out += [qk(0, m) for m in range(16)]
    out += [qk(1, m) for m in range(16)]
    out.append(wait('s_waitcnt vmcnt(16)\\n\\ts_barrier'))          # K(1) landed everywhere; all reads of slot 0 retired (lgkmcnt(0) above)
    out += [read_k(i, 1) for i in range(16)]
    out += emit(softmax_ops(0))
    return out


def body(J):
    out = [f'// generated by tools/gen_attn3.py -- iteration t, ring slot J = t & 3 = {J}']
    out.append('A3_TR(3)')
    out.append(wait('s_waitcnt vmcnt(8) lgkmcnt(0)', own=True))
    out.append('A3_TR(4)')
    out.append(wait('s_barrier'))
    out.append('A3_TR(0)')
    out.append('{ const uint64_t ksrc = k_src(t + 4), vsrc = v_src(t + 2); A3_KOFS(t + 4)')
    # phase A: S_A(t+1), O_A += V(t) P_A(t) | softmax of S_B(t) | V(t) fragments, DMA K(t+4) -> slot J
    sm = softmax_ops(1)
    cnt = valu_counts(True)
    k = 0
    out.append('// ---- phase A')
    for m in range(32):
        if m < 16:
            out.append(qk(0, m))
            if 'nolds' not in ABL:
                out.append(read_v(m, J))
        else:
            if m in (16, 20, 24, 28):
                out.append(wait(f's_waitcnt lgkmcnt({12 - (m - 16)})'))
            out.append(pv(0, m - 16))
            if m % 4 == 2 and 'nodma' not in ABL:
                out.append(dma('k', J, (m - 16) // 4))
        out += emit(sm[k:k + cnt[m]])
        k += cnt[m]
    assert k >= len(sm) or ABL
    out.append('A3_TR(1)')
    # phase B: S_B(t+1), O_B += V(t) P_B(t) | softmax of S_A(t+1) | DMA V(t+2) -> slot J+2, K(t+2) fragments from slot J+2
    sm = softmax_ops(0)
    cnt = valu_counts(False)
    k = 0
    out.append('// ---- phase B')
    out.append('if (__builtin_expect(KS != KS_pad && t + 2 == ntiles, 0)) { A3_MASK_S0 }')
    for m in range(32):
        if m < 16:
            out.append(qk(1, m))
            if m % 4 == 2 and 'nodma' not in ABL:
                out.append(dma('v', (J + 2) & 3, m // 4))
        else:
            out.append(pv(1, m - 16))
            if 'nolds' not in ABL:
                out.append(read_k(m - 16, (J + 2) & 3))
        out += emit(sm[k:k + cnt[m]])
        k += cnt[m]
    assert k >= len(sm) or ABL
    out.append('}')
    out.append('A3_TR(2)')
    return out


def final():
    out = ['// generated by tools/gen_attn3.py -- last tile t = ntiles - 1 (runtime ring slot: vs = (t & 3) * 16384)']
    out.append(wait('s_waitcnt vmcnt(8) lgkmcnt(0)\\n\\ts_barrier', own=True))
    out += [read_v(i, 0, 'vs') for i in range(16)]
    out.append('if (KS != KS_pad) { A3_MASK_S1 }')
    out += emit(softmax_ops(1))
    out.append(wait('s_waitcnt lgkmcnt(0)\\n\\ts_nop 3'))
    out += [pv(0, m) for m in range(16)]
    out += [pv(1, m) for m in range(16)]
    return out


def rescale(sl):
    X = 'AB'[sl]
    out = [f'// generated by tools/gen_attn3.py -- cold path: O^T of slab {X} *= alpha (accumulator file <-> VALU, explicit wait states)']
    lines = ['s_nop 15', 's_nop 15']          # MFMA write -> accumulator read
    for r in range(64):
        a = 64 * sl + r
        lines += [f'v_accvgpr_read_b32 %0, a{a}', 'v_mul_f32 %0, %0, %1', 's_nop 0', f'v_accvgpr_write_b32 a{a}, %0']
    lines += ['s_nop 7']                       # accumulator write -> MFMA SrcC
    out.append(f'#define A3_OSCALE_{X} ' + asm('\\n\\t'.join(lines), '"=&v"(rs_tmp)', '"v"(alpha)'))
    return out


def readout():
    out = ['// generated by tools/gen_attn3.py -- epilogue: one 32-row tile of O^T -> 16 VGPR scalars ox[0..15]']
    out.append('#define A3_DRAIN ' + asm('s_waitcnt vmcnt(0)\\n\\ts_nop 15\\n\\ts_nop 15', '', '', '"memory"'))
    for sl in range(2):
        for d in range(4):
            lo = 64 * sl + 16 * d
            text = '\\n\\t'.join(f'v_accvgpr_read_b32 %{r}, a{lo + r}' for r in range(16))
            outs = ', '.join(f'"=v"(ox[{r}])' for r in range(16))
            out.append(f'#define A3_READ_{sl}_{d} ' + asm(text, outs, ''))
    # the reverse (a segment that continues from handed-over partial results): ox[0..15] -> one 32-row tile of O^T
    for sl in range(2):
        for d in range(4):
            lo = 64 * sl + 16 * d
            text = '\\n\\t'.join(f'v_accvgpr_write_b32 a{lo + r}, %{r}' for r in range(16))
            ins = ', '.join(f'"v"(ox[{r}])' for r in range(16))
            out.append(f'#define A3_WRITE_{sl}_{d} ' + asm(text, '', ins))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='gen', help='directory under arcflow_amd/csrc (afx_attn3.hip includes A3_GEN/..., default gen)')
    ap.add_argument('--merge', action='store_true', help='one asm statement per MFMA gap for the VALU instructions')
    ap.add_argument('--ablate', default='', help='comma list of nodma, nolds, noadd, noexp, novalu: timing experiments, WRONG results')
    a = ap.parse_args()
    ABL.update(x for x in a.ablate.split(',') if x)
    global MERGE
    MERGE = a.merge
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'arcflow_amd', 'csrc', a.out)
    os.makedirs(root, exist_ok=True)
    files = {'a3_init.inc': decl(), 'a3_mask.inc': mask_ops(0) + mask_ops(1), 'a3_qload.inc': q_loads(), 'a3_tile0.inc': tile0(), 'a3_final.inc': final(),
             'a3_rescale.inc': rescale(0) + rescale(1), 'a3_readout.inc': readout(), 'a3_prologue_dma.inc': prologue_dma()}
    for J in range(4):
        files[f'a3_body{J}.inc'] = body(J)
    for f in os.listdir(root):
        if f.startswith('a3_') and f not in files:
            os.remove(os.path.join(root, f))
    for name, lines in files.items():
        with open(os.path.join(root, name), 'w') as f:
            f.write('\n'.join(lines) + '\n')
        print(name, len(lines), 'lines')


if __name__ == '__main__':
    main()
