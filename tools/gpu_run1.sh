cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_production_shape.py -m gpu -q -k "attention" 2>&1 | tail -8
timeout 200 python tools/attn_bench.py 2>&1 | tail -25
