for r in 1 2 3; do
  a=$(python bench.py --steps 100 --warmup 8 --no-roofline --no-secondary --cpu-clips 0 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['value'])")
  b=$(python bench.py --steps 100 --warmup 8 --no-roofline --no-secondary --cpu-clips 0 --no-graph 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['value'])")
  c=$(python bench.py --steps 20 --warmup 5 --no-roofline --no-secondary --cpu-clips 0 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['value'])")
  d=$(python bench.py --steps 20 --warmup 5 --no-roofline --no-secondary --cpu-clips 0 --no-graph 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['value'])")
  echo "round $r graph100 $a eager100 $b graph20 $c eager20 $d"
done
