#!/bin/bash
# One GPU-box session: the whole -m gpu suite (timed per test), per-kernel microbench, default bench line.  Logs -> gpurun_out/<tag>/
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -q -m gpu --durations=25 ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 600 python tools/microbench.py gemm attn elem > $OUT/micro.log 2>&1
cat $OUT/micro.log | grep -v amdgpu.ids
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
python tools/bench_brief.py < $OUT/bench.json
(python bench.py --gpus 2 --steps 1; echo "exit=$?") > $OUT/bench_gpus2.log 2>&1; tail -2 $OUT/bench_gpus2.log
