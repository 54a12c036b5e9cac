#!/bin/bash
# Interleaved A/B of env settings on the training step (BASELINE config 3, --loss simple = one pass): tools/ab_train.sh "<name>=<ENV ...>" ...
set -u
for r in 1 2 3; do
  for arm in "$@"; do
    name=${arm%%=*}; envs=${arm#*=}
    v=$(env $envs timeout -k 10 300 python tools/train_bench.py --steps 10 --warmup 3 --loss simple 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(round(j['value'],1), j.get('ms_per_step'))")
    echo "round $r $name [$envs] $v"
  done
done
