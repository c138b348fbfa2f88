#!/bin/bash
# Interleaved same-box A/B of several builds of the library:  tools/ab_libs.sh <tag> <rounds> <lib suffix ...>   ('' = the in-tree product library)
#   e.g. tools/ab_libs.sh r06b 3 '' _notail _r05      (arcflow_amd/lib/libarcflow_hip<suffix>.so; python -m arcflow_amd.build --variant <name> ...)
TAG=$1; ROUNDS=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
LIBDIR=$(pwd)/arcflow_amd/lib
for s in "$@"; do
  echo "=== microbench gemm, lib '$s'" | tee -a $OUT/micro.log
  if [ -n "$s" ]; then export ARCFLOW_HIP_LIB=$LIBDIR/libarcflow_hip$s.so; else unset ARCFLOW_HIP_LIB; fi
  timeout 600 python tools/microbench.py gemm 2>&1 | grep -v amdgpu.ids | tee -a $OUT/micro.log
done
for r in $(seq 1 $ROUNDS); do
  for s in "$@"; do
    if [ -n "$s" ]; then export ARCFLOW_HIP_LIB=$LIBDIR/libarcflow_hip$s.so; else unset ARCFLOW_HIP_LIB; fi
    echo -n "lib '$s'  "; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras ${AB_ARGS} 2>/dev/null | python tools/bench_brief.py
  done
done 2>&1 | tee $OUT/ab.log
