cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "shapes: 9216x3072 3072x3072 12288x3072 3072x12288 21504x3072 3072x15360 8192^3  (N x K, M = 4608)"
for rep in 1 2; do
for lib in "" _nt _stg_tm2 _stg_tn2 _stg_wg _stg_tm8; do
  printf "%-10s " "base$lib"; ARCFLOW_HIP_LIB=$PWD/arcflow_amd/lib/libarcflow_hip$lib.so timeout 200 python tools/gemm_shapes_time.py 2>&1 | tail -1
done
printf "%-10s " "hipblaslt"; TORCH=1 timeout 200 python tools/gemm_shapes_time.py 2>&1 | tail -1
done
