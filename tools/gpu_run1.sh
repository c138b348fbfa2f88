cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_inference_script.py tests/test_checkpoint.py tests/test_distill.py "tests/test_hip_kernels.py::test_linear_stream_k_tail" "tests/test_hip_kernels.py::test_linear_stream_k_gate_residual_inplace" -x -q -m gpu > gpurun_out/t_new.log 2>&1; tail -30 gpurun_out/t_new.log
