#!/bin/bash
# Quick GPU check after a kernel / engine change: the engine + kernel parity tests, then interleaved bench A/B runs.
#   tools/gpu_quick.sh <tag> [env-var-name] [microbench targets...]     (A/B of ENV=0 vs ENV=1 when a name is given)
TAG=${1:-q}; VAR=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_engine.py tests/test_production_shape.py tests/test_pipeline.py -q -m gpu -x ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
if [ -n "$1" ]; then timeout 600 python tools/microbench.py "$@" 2>&1 | grep -v amdgpu.ids | tee $OUT/micro.log; fi
for r in 1 2 3; do
  if [ -n "$VAR" ] && [ "$VAR" != "-" ]; then
    echo -n "$VAR=0 "; env $VAR=0 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/bench_brief.py
    echo -n "$VAR=1 "; env $VAR=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/bench_brief.py
  else
    timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/bench_brief.py
  fi
done 2>&1 | tee $OUT/ab.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.log 2>/dev/null
cd $GRAFT_REPO_ROOT
f=$(ls $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cut -c1-150 $f | head -12
