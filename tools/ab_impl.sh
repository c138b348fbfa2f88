#!/bin/bash
# interleaved bench.py A/B of two AFX_GEMM_IMPL values (+ parity tests of the candidate): tools/ab_impl.sh <base> <candidate>
A=$1; B=$2
AFX_GEMM_IMPL=$B timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_vae.py tests/test_hip_engine.py -x -q -m gpu 2>&1 | tail -2
AFX_GEMM_IMPL=$B python tools/microbench.py gemm 2>&1 | tail -9
for r in 1 2 3; do
  for i in $B $A; do
    echo -n "impl$i "; AFX_GEMM_IMPL=$i timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/bench_brief.py
  done
done
