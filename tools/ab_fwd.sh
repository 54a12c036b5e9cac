#!/bin/bash
# Interleaved A/B of env settings on the headline forward (100 timed steps, no roofline pass, no secondary): tools/ab_fwd.sh "<name>=<ENV ...>" ...
# extra bench flags through ABFLAGS (e.g. ABFLAGS="--arch resnet2p1d_18 --batch 8 --frames 32 --size 112")
set -u
for r in 1 2 3; do
  for arm in "$@"; do
    name=${arm%%=*}; envs=${arm#*=}
    v=$(env $envs timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-roofline --no-secondary --cpu-clips 0 ${ABFLAGS:-} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(round(j['value'],1), j.get('ms_per_step'))")
    echo "round $r $name [$envs] $v"
  done
done
