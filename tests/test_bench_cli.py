"""bench.py's launcher contract (CPU): `--gpus N` never silently degrades to one GPU (VERDICT r01 "What's missing" 1)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_single_gpu_runs_in_process():
    import bench
    args = bench.parse_args(['--steps', '3'])
    assert args.gpus == 1 and args.steps == 3 and args.warmup == 2
    assert bench.launcher_cmd(args, ['--steps', '3'], environ={}, device_count=1) is None


def test_multi_gpu_self_launch_command():
    import bench
    argv = ['--gpus', '4', '--steps', '7', '--warmup', '1', '--model', 'qwen']
    args = bench.parse_args(argv)
    cmd = bench.launcher_cmd(args, argv, environ={}, device_count=8)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == argv                      # every flag is forwarded to the ranks


def test_multi_gpu_request_on_a_small_box_fails_loudly():
    import bench
    argv = ['--gpus', '2']
    with pytest.raises(SystemExit) as e:
        bench.launcher_cmd(bench.parse_args(argv), argv, environ={}, device_count=1)
    assert '--gpus 2' in str(e.value) and '1 GPU' in str(e.value)


def test_rank_of_a_torchrun_launch_runs_in_process_and_checks_world_size():
    import bench
    argv = ['--gpus', '8']
    assert bench.launcher_cmd(bench.parse_args(argv), argv, environ={'WORLD_SIZE': '8', 'RANK': '3'}, device_count=8) is None
    with pytest.raises(SystemExit):
        bench.launcher_cmd(bench.parse_args(argv), argv, environ={'WORLD_SIZE': '2', 'RANK': '0'}, device_count=8)
    with pytest.raises(SystemExit):      # the driver's bare `python bench.py` under a stale WORLD_SIZE
        bench.launcher_cmd(bench.parse_args([]), [], environ={'WORLD_SIZE': '4'}, device_count=8)


def test_train_mode_defaults():
    import bench
    a = bench.parse_args(['--train', '--model', 'qwen'])
    assert a.train and a.steps == 2 and a.warmup == 1 and a.batch is None
