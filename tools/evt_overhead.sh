for r in 1 2 3; do
  a=$(python bench.py --steps 20 --warmup 5 --cpu-clips 0 --no-secondary 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['value'])")
  b=$(python bench.py --steps 20 --warmup 5 --cpu-clips 0 --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['value'])")
  c=$(python bench.py --steps 200 --warmup 5 --cpu-clips 0 --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['value'])")
  echo "round $r events20 $a plain20 $b plain200 $c"
done
