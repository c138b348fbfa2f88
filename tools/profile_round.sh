#!/bin/bash
# Evidence of a round on one MI355X box: rocprofv3 kernel tables (headline bench, distillation iteration, attention backward), PMC traffic
# passes of the headline bench (separate --pmc runs, kernel-trace only: gpurun refuses --pmc together with the other trace domains), the
# final bench lines and the --no-profile A/B of the headline.
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
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -- python $R/bench.py --train --steps 1 --warmup 1 > $O/train_under_rocprof.log 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_attn_bwd -- python $R/tools/attn_bwd_bench.py > $O/attn_bwd_under_rocprof.log 2>/dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_fwd -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_bwd -- python $R/tools/attn_bwd_bench.py > /dev/null 2>&1
cd $R
( echo '# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- bench.py (forward) / tools/attn_bwd_bench.py: mean per dispatch by (kernel, grid)'; python tools/pmc_by_grid.py "$O/mfma_fwd/*/*counter_collection.csv" "$O/mfma_bwd/*/*counter_collection.csv" ) > $O/pmc_mfma_busy.txt 2>&1
( echo '== FETCH_SIZE (KiB per dispatch, mean)'; python tools/pmc_summary.py $(ls $O/fetch/*/*counter_collection.csv | head -1) | grep -A1 'gemm_kernel\|attention_v3' ;
  echo '== WRITE_SIZE'; python tools/pmc_summary.py $(ls $O/write/*/*counter_collection.csv | head -1) | grep -A1 'gemm_kernel\|attention_v3' ) > $O/pmc_fetch_write_size.txt 2>&1
# profiles/traffic.json of THIS build (stamped with the kernel sources' sha; copy the printed file back: gpurun_out/<tag>/traffic.json)
python tools/update_traffic.py $(ls $O/fetch/*/*counter_collection.csv | head -1) $(ls $O/write/*/*counter_collection.csv | head -1) $TAG > $O/traffic_update.txt 2>&1; cp profiles/traffic.json $O/traffic.json
python tools/attn_bwd_bench.py > $O/attn_bwd_bench.txt 2>/dev/null
# the headline with and without the per-launch HIP events, interleaved
for i in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --profile-stride 1 2>/dev/null | python tools/bench_brief.py "events on every launch " >> $O/no_profile_ab.txt
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py "events on 1 launch in 8 " >> $O/no_profile_ab.txt
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-profile 2>/dev/null | python tools/bench_brief.py "events off             " >> $O/no_profile_ab.txt
done
# fp8 (configs[4]): where an fp8 forward goes with block-scaled activations (default) and with one scale per row + a quantisation pass per GEMM
cd /tmp
for v in 1 0; do
  AFX_FP8_MX=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp8_mx$v -- python $R/bench.py --fp8 --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
done
cd $R
( echo '# rocprofv3 --kernel-trace --stats -- python bench.py --fp8 --steps 4 --warmup 2 --no-profile   (block-scaled activations, the default)'; python tools/kstats_top.py $(ls $O/stats_fp8_mx1/*/*kernel_stats.csv | head -1) 12 ) > $O/kernel_stats_bench_flux_fp8_blockscaled.txt 2>&1
( echo '# AFX_FP8_MX=0 rocprofv3 --kernel-trace --stats -- python bench.py --fp8 --steps 4 --warmup 2 --no-profile   (one scale per row, a quantisation pass per GEMM)'; python tools/kstats_top.py $(ls $O/stats_fp8_mx0/*/*kernel_stats.csv | head -1) 12 ) > $O/kernel_stats_bench_flux_fp8_rowscaled.txt 2>&1
python tools/microbench.py gemm8 2>/dev/null | grep -v amdgpu > $O/fp8_gemm_microbench.txt
for i in 1 2; do
  python bench.py --fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py "fp8 block-scaled       " >> $O/fp8_forward_ab.txt
  AFX_FP8_MX=0 python bench.py --fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py "fp8 row-scaled         " >> $O/fp8_forward_ab.txt
  AFX_FP8_MX=0 AFX_FP8_V3=0 python bench.py --fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py "fp8 row-scaled, 8-phase" >> $O/fp8_forward_ab.txt
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py "bf16                   " >> $O/fp8_forward_ab.txt
done
# training iteration: one stream / text side stream / + weight-gradient stream (default)
for i in 1 2; do
  ARCFLOW_TRAIN_TXT_STREAM=0 ARCFLOW_TRAIN_WGRAD_STREAM=0 python bench.py --train --steps 2 --warmup 1 2>/dev/null | python tools/bench_brief.py "flux train, one stream     " >> $O/train_streams_ab.txt
  ARCFLOW_TRAIN_WGRAD_STREAM=0 python bench.py --train --steps 2 --warmup 1 2>/dev/null | python tools/bench_brief.py "flux train, text stream    " >> $O/train_streams_ab.txt
  python bench.py --train --steps 2 --warmup 1 2>/dev/null | python tools/bench_brief.py "flux train, three streams  " >> $O/train_streams_ab.txt
done
python bench.py > $O/bench_default_line.json 2>/dev/null
python bench.py --model qwen --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_qwen.json 2>/dev/null
python bench.py --streams 2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_flux_2streams.json 2>/dev/null
python bench.py --train --steps 2 --warmup 1 > $O/bench_train_flux.json 2>$O/bench_train_flux.err
python bench.py --train --model qwen --steps 2 --warmup 1 > $O/bench_train_qwen.json 2>$O/bench_train_qwen.err
python bench.py --train --model qwen --teacher-fp8 --student-fp8 --steps 2 --warmup 1 > $O/bench_train_qwen_fp8.json 2>$O/bench_train_qwen_fp8.err
for f in $O/bench_*.json; do echo $f; python tools/bench_brief.py < $f 2>/dev/null || head -c 400 $f; done
cat $O/no_profile_ab.txt $O/fp8_forward_ab.txt $O/train_streams_ab.txt $O/fp8_gemm_microbench.txt $O/kernel_stats_bench_flux_fp8_blockscaled.txt $O/kernel_stats_bench_flux_fp8_rowscaled.txt $O/attn_bwd_bench.txt $O/pmc_fetch_write_size.txt $O/pmc_mfma_busy.txt
ls $O/*/* | head -40
