cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
t0=$(date +%s)
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
t1=$(date +%s)
echo "default bench.py wall: $((t1-t0)) s"
tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().split('\n')[-1])
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('value','frac','achieved','ms_per_step','kind','cores','unit','max_mem_gb','dtype')}) for k,v in d.items() if k not in ('config','metric','data')})
PY
