cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
for impl in 1 0; do
AFX_ATTN_IMPL=$impl timeout 300 python bench.py --steps 12 --warmup 3 2>/dev/null | python tools/bench_brief.py "attn_impl=$impl" >> gpurun_out/ab_attn.log 2>&1
done; done
cat gpurun_out/ab_attn.log
