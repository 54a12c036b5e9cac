for r in 1 2 3; do
  for w in 5 60 200; do
    v=$(python bench.py --steps 20 --warmup $w --cpu-clips 0 --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['value'])")
    echo "round $r warmup $w steps 20: $v"
  done
done
