"""The ISA audits arcflow_amd/build.py runs after every build, on synthetic assembly (CPU): the accumulator file of gemm_kernel_v3 / gemm_kernel_v3s is
asm-owned (round 6) -- a compiler-generated instruction that names an accumulator register at or above the kernel's base must fail the build, the
kernel's own asm statements (between ASMSTART / ASMEND) and hipcc's parking below the base must not."""
import pytest

from arcflow_amd import build

GOOD = """
_ZN3afx14gemm_kernel_v3ILi8ELi8ELb0ELi0EEEvNS_9GemmBatchE: ; @kernel
	v_add_u32_e32 v1, v2, v3
	;;#ASMSTART
	v_mfma_f32_16x16x32_bf16 a[0:3], v[4:7], v[8:11], a[0:3]
	;;#ASMEND
	;;#ASMSTART
	v_accvgpr_read_b32 v12, a17
	;;#ASMEND
	buffer_store_dwordx4 v[12:15], v1, s[8:11], 0 offen
_ZN3afx15gemm_kernel_v3sILi4ELi4ELb0ELi0EEEvNS_9GemmBatchE: ; @kernel
	v_accvgpr_write_b32 a3, v1
	v_accvgpr_read_b32 v1, a31
	;;#ASMSTART
	v_mfma_f32_16x16x32_bf16 a[32:35], v[4:7], v[8:11], a[32:35]
	;;#ASMEND
_ZN3afx16gemm_kernel_v3f8ILi8ELi8ELb0EEEvNS_9GemmBatchE: ; @kernel (hipcc-owned accumulators: not audited)
	v_accvgpr_read_b32 v1, a200
"""
KERNELS = {'gemm_kernel_v3I': 0, 'gemm_kernel_v3sI': 32}


def _write(tmp_path, text):
    p = tmp_path / 'k.s'
    p.write_text(text)
    return str(p)


def test_accumulator_audit_accepts_asm_owned_use_and_parking_below_the_base(tmp_path):
    build.audit_acc_owned(_write(tmp_path, GOOD), KERNELS)


@pytest.mark.parametrize('bad', ['\tv_accvgpr_read_b32 v9, a40\n', '\tscratch_store_dwordx4 off, a[36:39], off offset:16\n', '\tv_accvgpr_mov_b32 a32, a2\n'])
def test_accumulator_audit_rejects_compiler_generated_access(tmp_path, bad):
    text = GOOD.replace('\tv_accvgpr_read_b32 v1, a31\n', '\tv_accvgpr_read_b32 v1, a31\n' + bad)
    with pytest.raises(RuntimeError, match='asm-owned accumulator'):
        build.audit_acc_owned(_write(tmp_path, text), KERNELS)
    big = GOOD.replace('\tv_add_u32_e32 v1, v2, v3\n', '\tv_add_u32_e32 v1, v2, v3\n\tv_accvgpr_write_b32 a0, v5\n')       # base 0: nothing of the file is hipcc's
    with pytest.raises(RuntimeError, match='asm-owned accumulator'):
        build.audit_acc_owned(_write(tmp_path, big), KERNELS)


def test_accumulator_audit_notices_a_renamed_kernel(tmp_path):
    with pytest.raises(RuntimeError, match='found no kernel'):
        build.audit_acc_owned(_write(tmp_path, GOOD), {'gemm_kernel_v4I': 0})
