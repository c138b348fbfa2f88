cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do echo -n "LORA_SPLITK=$v "; ARCFLOW_LORA_SPLITK=$v timeout 600 python bench.py --train --steps 2 --warmup 1 2>/dev/null | python tools/bench_brief.py; done
