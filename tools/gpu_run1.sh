cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python tools/train.py examples/flux_distill_2nfe.py --synthetic --iters 4 --work-dir /tmp/soak --cfg-options train_cfg.student_fp8=True train_cfg.teacher_fp8=True checkpoint_config.interval=2 > gpurun_out/r03z_train_cli_soak_fp8.log 2>&1
tail -8 gpurun_out/r03z_train_cli_soak_fp8.log | cut -c1-300
