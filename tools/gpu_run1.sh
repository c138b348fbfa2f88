cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/t_all.log 2>&1; tail -8 gpurun_out/t_all.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
