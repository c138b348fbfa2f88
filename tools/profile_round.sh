#!/bin/bash
# Evidence of a round on one MI355X box: rocprofv3 kernel table + PMC traffic passes of the headline bench, final bench lines.
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...   (copy what you want judged into profiles/)
set -x
TAG=${1:-pf}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.log 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
for tpw in 1 2 4; do
  AFX_STEP_TPW=$tpw rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_tpw$tpw -- python $R/tools/step_bench.py > $O/step_tpw$tpw.log 2>/dev/null
done
cd $R
python bench.py --steps 10 --warmup 2 > $O/bench_flux.json 2>/dev/null
python bench.py --model qwen --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_qwen.json 2>/dev/null
python bench.py --streams 2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_flux_2streams.json 2>/dev/null
python bench.py --train --steps 2 --warmup 1 > $O/bench_train_flux.json 2>$O/bench_train_flux.err
python bench.py --train --model qwen --steps 2 --warmup 1 > $O/bench_train_qwen.json 2>$O/bench_train_qwen.err
python bench.py --train --model qwen --teacher-fp8 --steps 2 --warmup 1 > $O/bench_train_qwen_fp8.json 2>$O/bench_train_qwen_fp8.err
for f in $O/bench_*.json; do echo $f; python tools/bench_brief.py < $f 2>/dev/null || head -c 400 $f; done
for t in 1 2 4; do f=$(ls $O/step_tpw$t/*/*kernel_stats.csv | head -1); grep arcflow_step $f | cut -c1-60,200-260; done
ls $O/*/* | head -30
