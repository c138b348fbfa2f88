set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  AFX_FP8_MX=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp8_mx$v -- python $R/bench.py --fp8 --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
done
cd $R
( echo '# rocprofv3 --kernel-trace --stats -- python bench.py --fp8 --steps 4 --warmup 2 --no-profile   (final build: LayerNorm operands row-scaled in the LayerNorm kernel, epilogue-produced operands block-scaled, attention writes fp8)'; python tools/kstats_top.py $(ls $O/stats_fp8_mx1/*/*kernel_stats.csv | head -1) 12 ) > $O/kernel_stats_bench_flux_fp8_final.txt 2>&1
( echo '# AFX_FP8_MX=0 rocprofv3 --kernel-trace --stats -- python bench.py --fp8 --steps 4 --warmup 2 --no-profile   (one scale per row, a quantisation pass per GEMM)'; python tools/kstats_top.py $(ls $O/stats_fp8_mx0/*/*kernel_stats.csv | head -1) 12 ) > $O/kernel_stats_bench_flux_fp8_rowscaled.txt 2>&1
for i in 1 2; do
  python bench.py --fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py "fp8 final (hybrid scales, no quantisation pass)  " >> $O/fp8_forward_ab.txt
  AFX_FP8_NORM_MX=1 python bench.py --fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py "fp8 block scales everywhere                      " >> $O/fp8_forward_ab.txt
  AFX_FP8_NORM_MX=1 AFX_FP8_ATTN_MX_OFF=1 python bench.py --fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py "  ... and the attention output quantised by a pass" >> $O/fp8_forward_ab.txt
  AFX_FP8_MX=0 python bench.py --fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py "fp8 row-scaled, a pass per GEMM                  " >> $O/fp8_forward_ab.txt
  AFX_FP8_MX=0 AFX_FP8_V3=0 python bench.py --fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py "fp8 row-scaled, 8-phase kernel (round 3)         " >> $O/fp8_forward_ab.txt
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_brief.py "bf16 (the headline mode)                         " >> $O/fp8_forward_ab.txt
done
python bench.py > $O/bench_default_line.json 2>/dev/null
python bench.py --train --model qwen --teacher-fp8 --student-fp8 --steps 2 --warmup 1 > $O/bench_train_qwen_fp8.json 2>/dev/null
python bench.py --model qwen --fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_qwen_fp8.json 2>/dev/null
cat $O/fp8_forward_ab.txt $O/kernel_stats_bench_flux_fp8_final.txt
for f in $O/bench_*.json; do echo $f; python tools/bench_brief.py < $f; done
