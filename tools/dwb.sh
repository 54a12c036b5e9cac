for cfg in "54 1 16 56 56" "108 1 16 28 28" "216 1 16 14 14" "432 1 16 7 7" "54 2 16 112 112" "108 2 16 56 56" "216 2 16 28 28" "432 2 16 14 14"; do
  echo "== $cfg"
  PASN_NO_DWMARCH=1 python tools/kbench.py dw $cfg 2>&1 | tail -1
  PASN_DWM_WT=2 python tools/kbench.py dw $cfg 2>&1 | tail -1
  PASN_DWM_WT=4 python tools/kbench.py dw $cfg 2>&1 | tail -1
done
