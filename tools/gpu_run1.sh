cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_vae.py -m gpu -q -x -k "softmax or decoder_vs or 1024" 2>&1 | grep -E "^E  |passed|failed" | head -20
for i in 1 2; do timeout 300 python tools/vae_bench.py 2>&1 | tail -2; done
