cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/attn3_ablate.log
for v in trace t_merge t_nodma t_mergenodma t_nolds t_noadd t_noexp t_novalu trace; do
echo "== $v" >> gpurun_out/attn3_ablate.log
ARCFLOW_HIP_LIB=$PWD/arcflow_amd/lib/libarcflow_hip_$v.so timeout 120 python tools/attn3_trace.py 2>&1 | grep "block 0 w0\|block 300 w0" >> gpurun_out/attn3_ablate.log
done
cat gpurun_out/attn3_ablate.log
