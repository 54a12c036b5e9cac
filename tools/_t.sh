for cfg in "54 1 16 56 56" "108 1 16 28 28" "216 1 16 14 14" "432 1 16 7 7"; do
  python tools/kbench.py dw $cfg 2>&1 | tail -1 | cut -c1-110
done
for r in 1 2; do python bench.py --steps 40 --warmup 5 --no-roofline --cpu-clips 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench (no roofline)', d['value'])"; done
