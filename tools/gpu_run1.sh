cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 300 python tools/vae_bench.py qwen 2>&1 | tail -1; done
for i in 1 2; do AFX_VAE_FOLD_UPSAMPLE=0 timeout 300 python tools/vae_bench.py qwen 2>&1 | tail -1; done
for i in 1 2; do timeout 300 python tools/vae_bench.py 2>&1 | tail -2; done
