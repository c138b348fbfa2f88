cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vae.py -m gpu -q -x 2>&1 | tail -8
for i in 1 2; do
AFX_GEMM_IMPL=2 timeout 300 python tools/vae_bench.py 2>&1 | tail -2
timeout 300 python tools/vae_bench.py 2>&1 | tail -2
done
