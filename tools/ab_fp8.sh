#!/bin/bash
# Interleaved same-box A/B of the fp8 paths over library builds:  tools/ab_fp8.sh <tag> <rounds> <lib suffix ...>   ('' = the product library)
TAG=$1; ROUNDS=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
LIBDIR=$(pwd)/arcflow_amd/lib
for s in "$@"; do
  if [ -n "$s" ]; then export ARCFLOW_HIP_LIB=$LIBDIR/libarcflow_hip$s.so; else unset ARCFLOW_HIP_LIB; fi
  echo "=== microbench gemm8, lib '$s'" | tee -a $OUT/micro8.log
  timeout 600 python tools/microbench.py gemm8 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tee -a $OUT/micro8.log
done
for r in $(seq 1 $ROUNDS); do
  for s in "$@"; do
    if [ -n "$s" ]; then export ARCFLOW_HIP_LIB=$LIBDIR/libarcflow_hip$s.so; else unset ARCFLOW_HIP_LIB; fi
    echo -n "fp8 forward, lib '$s'  "; timeout 300 python bench.py --fp8 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py
  done
done 2>&1 | tee $OUT/ab_fp8_forward.log
for r in 1 2; do
  for s in "$@"; do
    if [ -n "$s" ]; then export ARCFLOW_HIP_LIB=$LIBDIR/libarcflow_hip$s.so; else unset ARCFLOW_HIP_LIB; fi
    echo -n "qwen fp8 train, lib '$s'  "; timeout 900 python bench.py --train --model qwen --teacher-fp8 --student-fp8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('ms/iter=%.1f' % d['ms_per_step'])"
  done
done 2>&1 | tee $OUT/ab_fp8_train.log
