# Timing ablations of the fused residual-block launch (needs lib/libprotoasnet_amd_tuning.so: protoasnet_amd.build_extension(variant="tuning"))
#   bash tools/block_abl.sh 0 1 2 3 4 8 12 15     (bits: 1 stencil MFMAs, 2 frame DMA, 4 project phase, 8 expand phase; results are wrong when set)
R=${GRAFT_REPO_ROOT:-/root/repo}
export PASN_LIB_PATH=$R/protoasnet_amd/lib/libprotoasnet_amd_tuning.so
for A in "$@"; do
  if [ "$A" != "0" ]; then export PASN_BLOCK_ABL=$A; else unset PASN_BLOCK_ABL; fi
  echo "== PASN_BLOCK_ABL=$A"
  python3 $R/tools/block_bench.py 20
done
