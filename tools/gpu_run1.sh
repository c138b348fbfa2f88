# scratch runner for gpurun calls during development: gpurun -- 'bash tools/gpu_run1.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
