set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pf && mkdir -p $R/gpurun_out/pf
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pf/bench_under_rocprof.log 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pf/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pf/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
cd $R
python bench.py --steps 10 --warmup 2 > gpurun_out/pf/bench_final.json 2>/dev/null
python bench.py --model qwen --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/pf/bench_qwen.json 2>/dev/null
ls gpurun_out/pf/*/* | head
