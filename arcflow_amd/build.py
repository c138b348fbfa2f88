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
SOURCES = ['afx_gemm.hip', 'afx_attn.hip', 'afx_attn3.hip', 'afx_attn_bwd.hip', 'afx_attn_bwd3.hip', 'afx_elementwise.hip', 'afx_train.hip', 'afx_vae.hip', 'afx_text.hip', 'afx_tn.hip', 'afx_engine.hip']
HEADERS = ['afx_common.h', 'afx_kernels.h', 'afx_api_util.h', 'afx_attn_bwd3_kernel.inc', os.path.join('..', '..', 'include', 'arcflow_hip.h')]
HEADERS += [os.path.join('gen', f) for f in sorted(os.listdir(os.path.join(CSRC, 'gen'))) if f.endswith('.inc')]
# sources whose kernels OWN registers by literal name (tools/gen_attn3.py): their ISA is audited after every build
ASM_OWNED = {'afx_attn3.hip': ('attention_v3_kernel', 96), 'afx_attn_bwd3.hip': ('attn_bwd_dkv3_kernel', 60, 'attn_bwd_dq3_kernel', 'attn_bwd_fused3_kernel')}
# kernels whose accumulators are written by inline-asm MFMAs the compiler cannot see into: a register spill there may store an accumulator straight
# behind the MFMA that is still writing it (happened to gemm_kernel_v3f8 once its epilogue grew) -- they must not use scratch at all
NO_SCRATCH = {'afx_gemm.hip': ['gemm_kernel_v3f8', 'gemm_kernel_v3ILi8ELi8ELb0ELi0E', 'gemm_kernel_v3ILi8ELi7ELb0ELi0E', 'gemm_kernel_v3ILi7ELi8ELb0ELi0E',
                               'gemm_kernel_v3ILi9ELi6ELb0ELi0E', 'gemm_kernel_v3sILi4ELi4ELb0ELi0E']}
# gemm_kernel_v3 / gemm_kernel_v3s (round 6): the accumulator file is asm-owned (literal register names in every MFMA and every epilogue read, afx_gemm.hip
# v3_mfma_lit / AccLit) and `amdgpu_num_vgpr` confines hipcc to the arch VGPRs: no compiler-generated instruction may touch an accumulator register
ACC_OWNED = {'afx_gemm.hip': {'gemm_kernel_v3I': 0, 'gemm_kernel_v3sI': 32, 'gemm_kernel_v3f8I': 0}}        # kernel -> first asm-owned accumulator register (hipcc may use the ones below)



def audit_no_scratch(asm_path: str, kernel_substrs) -> None:
    """Every kernel of the file whose mangled name contains one of `kernel_substrs` must have private_segment_fixed_size 0."""
    import re
    text = open(asm_path).read()
    seen = {k: 0 for k in kernel_substrs}
    for m in re.finditer(r'\.amdhsa_kernel (\S+)', text):
        name = m.group(1)
        hits = [k for k in kernel_substrs if k in name]
        if not hits:
            continue
        for k in hits:
            seen[k] += 1
        size = re.search(r'private_segment_fixed_size\s+(\d+)', text[m.start():m.start() + 4000])
        if size is None or int(size.group(1)) != 0:
            raise RuntimeError(f'{asm_path}: kernel {name} uses {size.group(1) if size else "?"} bytes of scratch: a spill inside an inline-asm MFMA stream '
                               f'is not hazard-checked by hipcc -- reduce its register pressure')
    missing = [k for k, n in seen.items() if n == 0]
    if missing:       # a listed instantiation that no longer exists (a changed template signature) would silently drop out of the audit
        raise RuntimeError(f'{asm_path}: the scratch audit found no kernel matching {missing}: update NO_SCRATCH in arcflow_amd/build.py')


def audit_acc_owned(asm_path: str, kernels) -> None:
    """Kernels whose accumulator file is asm-owned from register `base` up (kernels: name substring -> base): outside ASMSTART / ASMEND no instruction may name
    an accumulator register a<n> / a[n:m] with m >= base (hipcc is free to park its own values below base)."""
    import re
    reg = re.compile(r'\ba(\d+)\b|\ba\[\d+:(\d+)\]')
    base = None
    in_asm = False
    seen = {k: 0 for k in kernels}
    bad = []
    with open(asm_path) as f:
        for n, line in enumerate(f, 1):
            t = line.strip()
            label = t.split(';')[0].strip()
            if label.endswith(':') and not label.startswith('.L') and label:
                hits = [k for k in kernels if k in label]
                base = kernels[hits[0]] if hits else None
                for k in hits:
                    seen[k] += 1
            if base is None or not t:
                continue
            if 'ASMSTART' in t:
                in_asm = True
            elif 'ASMEND' in t:
                in_asm = False
            elif not in_asm and t[0] not in ';.':
                for m in reg.finditer(t.split(';')[0]):
                    if int(m.group(1) or m.group(2)) >= base:
                        bad.append(f'{n}: {t}')
                        break
    missing = [k for k, c in seen.items() if c == 0]
    if missing:
        raise RuntimeError(f'{asm_path}: the accumulator audit found no kernel matching {missing}: update ACC_OWNED in arcflow_amd/build.py')
    if bad:
        raise RuntimeError(f'{asm_path}: compiler-generated code touches the asm-owned accumulator registers ({len(bad)} instructions):\n' + '\n'.join(bad[:20]))


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


def audit_asm_owned(asm_path: str, kernel_substr: str, vgpr_limit: int = 96, *more_kernels) -> None:
    """A kernel whose registers are asm-owned (literal names in the asm text: v[vgpr_limit:255] and the whole accumulator file) must
    contain no compiler-generated instruction that touches them and no scratch access: hipcc cannot know they are in use, and
    a wrong amdgpu_num_vgpr ceiling was exceeded once (silent corruption), and past the right one hipcc spills into accumulator registers.
    Raises on a violation."""
    import re
    for other in more_kernels:
        audit_asm_owned(asm_path, other, vgpr_limit)
    reg = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]|\b(a)\d+\b|\b(a)\[\d+:\d+\]')
    in_kernel = in_asm = False
    seen = False
    bad = []
    with open(asm_path) as f:
        for n, line in enumerate(f, 1):
            t = line.strip()
            label = t.split(';')[0].strip()          # "_ZN3afx...E:      ; @_ZN3afx..."
            if label.endswith(':') and not label.startswith('.L') and label:
                in_kernel = kernel_substr in label
                seen = seen or in_kernel
            if not in_kernel or not t:
                continue
            if 'ASMSTART' in t:
                in_asm = True
            elif 'ASMEND' in t:
                in_asm = False
            elif not in_asm and t[0] not in ';.':
                code = t.split(';')[0]
                if t.startswith('v_accvgpr') or t.startswith('scratch_') or t.startswith('buffer_') and 'offen' in t:
                    bad.append(f'{n}: {t}')
                    continue
                for m in reg.finditer(code):
                    if m.group(4) or m.group(5) or int(m.group(1) or m.group(3)) >= vgpr_limit:
                        bad.append(f'{n}: {t}')
                        break
    if not seen:
        raise RuntimeError(f'{asm_path}: kernel {kernel_substr} not found by the register audit')
    if bad:
        raise RuntimeError(f'{asm_path}: compiler-generated code touches asm-owned registers / spills in {kernel_substr} '
                           f'({len(bad)} instructions):\n' + '\n'.join(bad[:20]))


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
        if src in ASM_OWNED or src in NO_SCRATCH:
            cmd += ['-save-temps=obj', '-Wno-inline-asm']      # keeps the .s next to the object for the audit below
        if verbose:
            print('[arcflow_amd.build]', ' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{log}')
        if verbose and log.strip():
            print(log)
        if src in ASM_OWNED:
            audit_asm_owned(os.path.join(objdir, src.replace('.hip', '-hip-amdgcn-amd-amdhsa-gfx950.s')), *ASM_OWNED[src])
        if src in NO_SCRATCH:
            audit_no_scratch(os.path.join(objdir, src.replace('.hip', '-hip-amdgcn-amd-amdhsa-gfx950.s')), NO_SCRATCH[src])
        if src in ACC_OWNED:
            audit_acc_owned(os.path.join(objdir, src.replace('.hip', '-hip-amdgcn-amd-amdhsa-gfx950.s')), ACC_OWNED[src])
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', out]
    if verbose:
        print('[arcflow_amd.build]', ' '.join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}')
    return out


def build_variant(name: str, extra_flags, sources) -> str:
    """lib/libarcflow_hip_<name>.so = the standard objects with `sources` recompiled under `extra_flags` (trace / A-B builds that
    travel next to the product library; select with ARCFLOW_HIP_LIB=<path>)."""
    build(verbose=False)
    objdir = os.path.join(LIBDIR, 'obj')
    vdir = os.path.join(LIBDIR, 'obj_' + name)
    os.makedirs(vdir, exist_ok=True)
    hipcc = _hipcc()
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-function', '-Wno-inline-asm', *extra_flags]
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        if src in sources:
            obj = os.path.join(vdir, src.replace('.hip', '.o'))
            r = subprocess.run([hipcc, *flags, '-c', os.path.join(CSRC, src), '-o', obj, '-save-temps=obj'], stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError(f'hipcc failed on {src} ({name}):\n{r.stdout}')
            if src in ASM_OWNED:
                audit_asm_owned(os.path.join(vdir, src.replace('.hip', '-hip-amdgcn-amd-amdhsa-gfx950.s')), *ASM_OWNED[src])
            if src in ACC_OWNED:
                audit_acc_owned(os.path.join(vdir, src.replace('.hip', '-hip-amdgcn-amd-amdhsa-gfx950.s')), ACC_OWNED[src])
        objs.append(obj)
    out = os.path.join(LIBDIR, f'libarcflow_hip_{name}.so')
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}')
    return out


if __name__ == '__main__':
    if '--variant' in sys.argv:         # python -m arcflow_amd.build --variant trace -DAFX_ATTN_TRACE -- afx_attn3.hip
        i = sys.argv.index('--variant')
        j = sys.argv.index('--')
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:j], sys.argv[j + 1:]))
    else:
        print(build(force='--force' in sys.argv))
