# phase costs of the matrix-core stencil by ablation (results are wrong when a bit is set): 1 no MFMA block, 2 no DMA, 4 no epilogue,
# 8 no wait + barrier
for cfg in "54 1 16 56 56" "216 1 16 14 14"; do
  for a in 0 1 2 4 8 3 5 6 7 15 14; do
    PASN_DWMFMA=1 PASN_DWMFMA_ABL=$a python tools/kbench.py dw $cfg 2>&1 | tail -1 | sed "s/$/ ABL=$a/"
  done
done
