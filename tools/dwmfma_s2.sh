# stride-2 depthwise stencils of X3D-S (N = 32, input planes 112 / 56 / 28): VALU kernel vs the matrix-core stencil, cost model's split
# and forced (T chunk, units per block) splits.  usage (GPU box): bash tools/dwmfma_s2.sh > gpurun_out/dwmfma_s2.txt
for cfg in "54 2 16 112 112" "108 2 16 56 56" "216 2 16 28 28"; do
  python tools/kbench.py dw $cfg 2>&1 | tail -1
  PASN_DWMFMA_S2=1 python tools/kbench.py dw $cfg 2>&1 | tail -1
  for sp in "16 1" "16 2" "16 4" "8 2" "8 4" "4 4"; do
    set -- $sp
    PASN_DWMFMA_S2=1 PASN_DWMFMA_TC=$1 PASN_DWMFMA_UPB=$2 python tools/kbench.py dw $cfg 2>&1 | tail -1 | sed "s/$/ TC=$1 UPB=$2/"
  done
done
