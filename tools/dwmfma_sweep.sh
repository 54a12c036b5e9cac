# depthwise stencil A/B over the stride-1 X3D-S layer shapes (N = 32): round 1's VALU kernel (the default) vs the opt-in matrix-core
# stencil with the cost model's split and forced (T chunk, units per block) splits
# usage (GPU box): bash tools/dwmfma_sweep.sh > gpurun_out/dwmfma_sweep.txt
for cfg in "54 1 16 56 56" "108 1 16 28 28" "216 1 16 14 14" "432 1 16 7 7"; do
  python tools/kbench.py dw $cfg 2>&1 | tail -1
  PASN_DWMFMA=1 python tools/kbench.py dw $cfg 2>&1 | tail -1
  for sp in "16 1" "16 2" "16 4" "16 8" "8 1" "8 2" "8 4" "4 2"; do
    set -- $sp
    PASN_DWMFMA=1 PASN_DWMFMA_TC=$1 PASN_DWMFMA_UPB=$2 python tools/kbench.py dw $cfg 2>&1 | tail -1 | sed "s/$/ TC=$1 UPB=$2/"
  done
done
