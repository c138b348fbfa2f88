#!/bin/bash
# interleaved bench.py A/B: the in-tree library vs a baseline build (arcflow_amd/lib/libarcflow_hip_base.so)
BASE=$(pwd)/arcflow_amd/lib/libarcflow_hip_base.so
for r in 1 2 3; do
  echo -n "new  "; timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ${AB_ARGS} 2>/dev/null | python tools/bench_brief.py
  echo -n "base "; ARCFLOW_HIP_LIB=$BASE timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ${AB_ARGS} 2>/dev/null | python tools/bench_brief.py
done
