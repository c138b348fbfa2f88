cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "linear and not race" > gpurun_out/t_gemm.log 2>&1; tail -5 gpurun_out/t_gemm.log
timeout 300 python tools/gemm_tile_ab.py > gpurun_out/gemm_tile_ab.log 2>&1; cat gpurun_out/gemm_tile_ab.log
for i in 1 2; do for pen in 1e9 1.03; do
AFX_GEMM_PEN224=$pen timeout 300 python bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python tools/bench_brief.py "pen224=$pen" >> gpurun_out/ab_224.log 2>&1
done; done; cat gpurun_out/ab_224.log
