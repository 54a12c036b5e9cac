#!/bin/bash
# Interleaved A/B on the R(2+1)D-18 forward (8 x 32 x 112 x 112): tools/ab_r2p1d.sh "<name>=<ENV ...>" ...
set -u
for r in 1 2 3; do
  for arm in "$@"; do
    name=${arm%%=*}; envs=${arm#*=}
    v=$(env $envs timeout -k 10 200 python bench.py --arch resnet2p1d_18 --batch 8 --frames 32 --size 112 --prototypes 40 --classes 4 --steps 30 --warmup 8 --no-roofline --no-secondary --cpu-clips 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(round(j['value'],1), j['ms_per_step'])")
    echo "round $r $name [$envs] $v"
  done
done
