cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/attn_dbg.log
: > $L
timeout 300 python -c "import torch; print(torch.cuda.get_device_name(0))" >> $L 2>&1
for n in 1 2 3 4 5 6 7 0; do
  AFX_ATTN3_DBG=$n timeout 45 python tools/attn_dbg.py >> $L 2>&1; echo "dbg $n exit $?" >> $L
done
S=256 H=2 timeout 45 python tools/attn_dbg.py >> $L 2>&1; echo "S256 exit $?" >> $L
cat $L
if grep -q "dbg 0 exit 0" $L; then
  timeout 600 python tools/attn_bench.py > gpurun_out/attn_bench.log 2>&1; echo "exit $?" >> gpurun_out/attn_bench.log
  tail -45 gpurun_out/attn_bench.log
fi
