"""Build libarcflow_hip.so (the C-ABI engine, include/arcflow_hip.h) for gfx950 with hipcc.

In-tree build: the shared object lands in arcflow_amd/lib/ so that it travels with the source
snapshot to the GPU box.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIBNAME = 'libarcflow_hip.so'
SOURCES = ['afx_gemm.hip', 'afx_attn.hip', 'afx_attn3.hip', 'afx_attn_bwd.hip', 'afx_elementwise.hip', 'afx_train.hip', 'afx_vae.hip', 'afx_text.hip', 'afx_engine.hip']
HEADERS = ['afx_common.h', 'afx_kernels.h', 'afx_api_util.h', os.path.join('..', '..', 'include', 'arcflow_hip.h')]


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or install ROCm under /opt/rocm)')


def needs_build() -> bool:
    out = lib_path()
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path."""
    out = lib_path()
    if not force and not needs_build():
        return out
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
    flags += os.environ.get('AFX_EXTRA_FLAGS', '').split()          # e.g. -DAFX_ATTN_TRACE for tools/attn_trace.py
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(obj)
        cmd = [hipcc, *flags, '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print('[arcflow_amd.build]', ' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{log}')
        if verbose and log.strip():
            print(log)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', out]
    if verbose:
        print('[arcflow_amd.build]', ' '.join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}')
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
