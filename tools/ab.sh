#!/bin/bash
# interleaved A/B of two GEMM implementations on the real forward (bench.py) + linear parity tests for the candidate
# usage: tools/ab.sh <impl_a> <impl_b>
A=$1; B=$2
AFX_GEMM_IMPL=$B timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "linear or gemm or conv" 2>&1 | tail -2
for r in 1 2; do
  for i in $B $A; do
    echo -n "impl$i "; AFX_GEMM_IMPL=$i timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/bench_brief.py
  done
done
