#!/bin/bash
# Interleaved same-box A/B of one distillation iteration under an environment switch:  tools/ab_train.sh <tag> <rounds> <model> VAR=a VAR=b ...
#   e.g. tools/ab_train.sh r06e 2 flux AFX_TN_SPLIT=0 AFX_TN_SPLIT=-1
TAG=$1; ROUNDS=$2; MODEL=$3; shift; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
EXTRA=""; [ "$MODEL" = "qwen" ] && EXTRA="--teacher-fp8 --student-fp8"
for r in $(seq 1 $ROUNDS); do
  for kv in "$@"; do
    echo -n "$kv  "; env $kv timeout 900 python bench.py --train --model $MODEL $EXTRA --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('ms/iter=%.1f' % d['ms_per_step'], d.get('metric', '')[:60])"
  done
done 2>&1 | tee $OUT/ab_train_$MODEL.log
