#!/bin/bash
# Interleaved end-to-end A/B of env settings on one box: tools/ab_bench.sh "<name>=<ENV ...>" ...   (3 rounds, bench.py --no-roofline --cpu-clips 0)
# prints clips/s per arm per round.  Example: tools/ab_bench.sh "base=PASN_WS=0" "ws=PASN_WS=1"
set -u
STEPS=${STEPS:-30}
for r in 1 2 3; do
  for arm in "$@"; do
    name=${arm%%=*}; envs=${arm#*=}
    v=$(env $envs timeout -k 10 200 python bench.py --steps $STEPS --warmup 8 --no-roofline --no-secondary --cpu-clips 0 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['value'],1))")
    echo "round $r $name [$envs] $v"
  done
done
