# tuning sweep for the march kernel: forced WT x Tc over the X3D-S depthwise layer shapes, then the cost model's pick
for cfg in "54 1 16 56 56" "108 1 16 28 28" "216 1 16 14 14" "432 1 16 7 7" "54 2 16 112 112" "108 2 16 56 56" "216 2 16 28 28" "432 2 16 14 14"; do
  for wt in 2 3; do for tc in 16 8; do
    PASN_DWM_WT=$wt PASN_DWM_TC=$tc python tools/kbench.py dw $cfg 2>&1 | tail -1
  done; done
  python tools/kbench.py dw $cfg 2>&1 | tail -1
done
