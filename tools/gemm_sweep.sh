#!/bin/bash
# A/B of the GEMM kernels / tile shapes with tools/gemm_trace.hip (200 back-to-back launches: the clock governor has settled)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAFX_GEMM_TRACE=${TRACE_LEVEL:-2} -I arcflow_amd/csrc tools/gemm_trace.hip -o /tmp/gt 2>/dev/null
for shape in ${SHAPES:-"3072 3072" "9216 3072" "12288 3072" "21504 3072" "3072 12288" "3072 15360"}; do
  set -- $shape
  for mode in "2 0" "3 1" "3 2"; do
    set -- $shape $mode
    echo "== N=$1 K=$2 impl=$3 tile=$4"
    TRACE_K=$2 AFX_GEMM_IMPL=$3 AFX_GEMM_TILE=$4 /tmp/gt $1 x y | grep "TF\|w0\|MHz" | head -${LINES_OUT:-2}
  done
done
