cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03z > gpurun_out/r03z_profile_round.log 2>&1
tail -5 gpurun_out/r03z_profile_round.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --train --model qwen --teacher-fp8 --student-fp8 --steps 2 --warmup 1 > gpurun_out/r03z/bench_train_qwen_fp8_both.json 2>/dev/null
python bench.py > gpurun_out/r03z/bench_default_line.json 2>/dev/null
for i in 1 2; do python tools/vae_bench.py 2>&1 | tail -2; done > gpurun_out/r03z/vae_bench.txt
cat gpurun_out/r03z/vae_bench.txt
