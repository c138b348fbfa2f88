"""The hand-placed instruction streams are GENERATED files that are committed (the build does not run the generators): these CPU tests keep them in step with their
generators and check the invariants the counted waits rest on."""
import filecmp
import importlib.util
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, 'arcflow_amd', 'csrc', 'gen')


def _run(tool, args, out):
    # the generators write under arcflow_amd/csrc/<--out>: a scratch directory name, removed afterwards
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', tool), '--out', out, *args], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return os.path.join(ROOT, 'arcflow_amd', 'csrc', out)


@pytest.mark.parametrize('tool,args,prefix', [('gen_attn3.py', [], 'a3_'), ('gen_attn_bwd3.py', [], 'b3_'), ('gen_attn_bwd3.py', ['--mode', 'dq'], 'q3_')])
def test_committed_streams_match_their_generators(tool, args, prefix):
    out = f'_gen_check_{prefix}{os.getpid()}'
    d = _run(tool, args, out)
    try:
        fresh = sorted(f for f in os.listdir(d) if f.startswith(prefix))
        committed = sorted(f for f in os.listdir(GEN) if f.startswith(prefix))
        assert fresh == committed
        for f in fresh:
            assert filecmp.cmp(os.path.join(d, f), os.path.join(GEN, f), shallow=False), f'{f}: regenerate with tools/{tool}'
    finally:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)


def _phase_ops(path):
    """(kind, text) of every statement of a generated phase: 'lds' for ds_read*, 'dma' for global_load_lds*, 'mfma', 'valu', 'wait'"""
    ops = []
    for line in open(path):
        m = re.search(r'asm volatile\("([^"]*)"', line)
        if not m:
            continue
        t = m.group(1)
        kind = ('lds' if t.startswith('ds_read') else 'dma' if 'global_load_lds' in t else 'mfma' if t.startswith('v_mfma') else
                'wait' if t.startswith('s_waitcnt') or t.startswith('s_barrier') else 'valu')
        ops.append((kind, t))
    return ops


@pytest.mark.parametrize('prefix,n_lds,n_dma,n_mfma,n_valu', [('b3', 56, 5, 32, 64), ('q3', 32, 4, 24, 56)])
def test_every_phase_of_the_backward_issues_the_same_memory_sequence(prefix, n_lds, n_dma, n_mfma, n_valu):
    """The counted waits (lgkmcnt / vmcnt are in-order counters) are computed by a queue model that assumes EVERY phase -- the peeled first two included -- issues the
    same sequence of LDS reads and DMA pieces; the steady-state phases also carry the full MFMA / VALU work."""
    def dest(t):
        bank, lo, hi = re.match(r'\S+ ([av])\[(\d+):(\d+)\]', t).groups()
        lo, hi = int(lo), int(hi)
        if bank == 'v' and 96 <= lo < 128:       # the two S / dP register sets alternate with the phase's parity (the -delta reads land in the NEXT phase's dP accumulator)
            lo, hi = lo - 32, hi - 32
        return f'{bank}[{lo}:{hi}]'

    def mem_seq(ops):
        # destination registers + kind: the ring-slot immediates differ from phase to phase, the sequence of destinations must not
        return [(k, dest(t) if k == 'lds' else '') for k, t in ops if k in ('lds', 'dma')]
    ref = None
    for name in ['p0', 'p1'] + [f'body{j}' for j in range(8)]:
        ops = _phase_ops(os.path.join(GEN, f'{prefix}_{name}.inc'))
        seq = mem_seq(ops)
        assert sum(k == 'lds' for k, _ in seq) == n_lds and sum(k == 'dma' for k, _ in seq) == n_dma, name
        if ref is None:
            ref = seq
        assert seq == ref, name
        if name.startswith('body'):
            assert sum(k == 'mfma' for k, _ in ops) == n_mfma and sum(k == 'valu' for k, _ in ops) == n_valu, name
        # a lgkmcnt wait never asks for more than the 4-bit counter holds
        for k, t in ops:
            for c in re.findall(r'lgkmcnt\((\d+)\)', t):
                assert int(c) <= 15


def test_backward_queue_model_orders_consumers_behind_their_reads():
    """Replay a steady-state phase pair through an independent in-order model: when an MFMA or a VALU instruction reads a register an LDS read writes, the waits issued
    since that read must have retired it (lgkmcnt(n): at most n reads outstanding)."""
    for prefix in ('b3', 'q3'):
        outstanding, pending = [], {}       # queue of reads in flight (destination register sets), register -> still in flight
        for name in ('body2', 'body3', 'body4'):
            for kind, t in _phase_ops(os.path.join(GEN, f'{prefix}_{name}.inc')):
                if kind == 'lds':
                    lo, hi = map(int, re.match(r'\S+ [av]\[(\d+):(\d+)\]', t).groups())
                    regs = {(t.split()[1][0], r) for r in range(lo, hi + 1)}
                    outstanding.append(regs)
                elif kind == 'wait':
                    m = re.search(r'lgkmcnt\((\d+)\)', t)
                    if m:
                        outstanding = outstanding[len(outstanding) - int(m.group(1)):] if int(m.group(1)) < len(outstanding) else outstanding
                elif kind in ('mfma', 'valu') and name != 'body2':     # (body2 warms the model up: its first consumers read what the phase before it loaded)
                    ops_txt = t.split(None, 1)[1]
                    srcs = ops_txt.split(',')[1:]                       # everything but the destination
                    used = set()
                    for s in srcs:
                        for bank, a, b in re.findall(r'([av])\[(\d+):(\d+)\]', s):
                            used |= {(bank, r) for r in range(int(a), int(b) + 1)}
                        for bank, a in re.findall(r'(?<![\[\d:])([av])(\d+)\b', s):
                            used.add((bank, int(a)))
                    for regs in outstanding:
                        assert not (regs & used), f'{prefix} {name}: {t} reads a register whose ds_read may still be in flight'
