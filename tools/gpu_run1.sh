cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_distill.py -m gpu -q -x -k "rccl" 2>&1 | tail -5
ARCFLOW_DP_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --train --steps 1 --warmup 1 2>&1 | tail -3 | cut -c1-1200
