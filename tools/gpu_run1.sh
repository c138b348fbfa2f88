cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2; do for g in 6 3 9 18; do echo -n "GROUP_M=$g "; AFX_GEMM_GROUP_M=$g timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py; done; done
