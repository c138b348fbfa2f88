cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_production_shape.py -m gpu -q -x -k "attention" 2>&1 | tail -3
for lib in libarcflow_hip_base.so libarcflow_hip.so libarcflow_hip_base.so libarcflow_hip.so; do
echo "== $lib"; ARCFLOW_HIP_LIB=$PWD/arcflow_amd/lib/$lib timeout 200 python tools/attn_bench.py 2>&1 | grep -E "impl 0 B=1 S=4608 H=24:|determinism"
done
