# depthwise stencil A/B over the X3D-S layer shapes: round-1 kernel (PASN_DWM2=0) vs every dwmarch2 instance (PASN_DWM2=CH,WT)
# usage (GPU box): bash tools/dw2_sweep.sh > gpurun_out/dw2_sweep.txt
for cfg in "54 1 16 56 56" "108 1 16 28 28" "216 1 16 14 14" "432 1 16 7 7"; do
  for v in 0 "8,3" "8,2" "4,4" "4,6" "4,3" ""; do
    PASN_DWM2=$v python tools/kbench.py dw $cfg 2>&1 | tail -1
  done
done
for cfg in "54 2 16 112 112" "108 2 16 56 56" "216 2 16 28 28" "432 2 16 14 14"; do
  for v in 0 "8,2" "4,3" "4,2" ""; do
    PASN_DWM2=$v python tools/kbench.py dw $cfg 2>&1 | tail -1
  done
done
