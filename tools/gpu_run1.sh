cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/full_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 >> gpurun_out/full_tests.log
cat gpurun_out/full_tests.log
