# depthwise micro-benchmark over the X3D-S layer shapes; extra env (e.g. PASN_DWM_TC=4) applies to every run
for cfg in "54 1 16 56 56" "108 1 16 28 28" "216 1 16 14 14" "432 1 16 7 7" "54 2 16 112 112" "108 2 16 56 56" "216 2 16 28 28" "432 2 16 14 14"; do
  python tools/kbench.py dw $cfg 2>&1 | tail -1
done
