cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_engine.py tests/test_production_shape.py -m gpu -q -x -k "norm or forward or engine" 2>&1 | tail -4
for r in 0 2 4 0 2 4; do echo "AFX_NM_ROWS=$r"; AFX_NM_ROWS=$r timeout 200 python tools/microbench.py elem 2>&1 | grep norm_modulate; done
for r in 0 4 2 0 4 2; do echo "AFX_NM_ROWS=$r"; AFX_NM_ROWS=$r timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py; done
