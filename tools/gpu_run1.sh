cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_engine.py tests/test_production_shape.py -m gpu -q -x -k "linear or gemm or qwen or forward or engine" 2>&1 | tail -4
for p in 1e9 1.03 1e9 1.03; do echo "PEN_QK224=$p"; AFX_GEMM_PEN_QK224=$p timeout 400 python bench.py --model qwen --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/bench_brief.py; done
