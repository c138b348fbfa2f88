cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_vae.py -m gpu -q -x -k "conv3x3 or decoder_vs" 2>&1 | grep -E "^E  |passed|failed" | head -20
for i in 1 2; do
AFX_VAE_CONV_STATS=0 timeout 300 python tools/vae_bench.py 2>&1 | tail -2
timeout 300 python tools/vae_bench.py 2>&1 | tail -2
done
rm -rf gpurun_out/vae_kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vae_kt -- python tools/vae_bench.py > gpurun_out/vae_bench.log 2>&1
python tools/kernel_trace_by_grid.py gpurun_out/vae_kt | cut -c1-150 | head -12
