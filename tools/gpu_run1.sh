cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in "--model qwen --teacher-fp8" "--model qwen --teacher-fp8 --student-fp8" "" "--student-fp8 --teacher-fp8"; do
  echo "== $f"
  timeout 900 python bench.py --train $f --steps 2 --warmup 1 2>gpurun_out/err.log | python tools/bench_brief.py || tail -5 gpurun_out/err.log
done
