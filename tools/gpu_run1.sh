cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_train_kernels.py tests/test_distill.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python bench.py --train --steps 2 --warmup 1 2>/dev/null | python tools/bench_brief.py
