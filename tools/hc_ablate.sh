# chain-head kernel time under its timing ablations (PASN_HC_ABL bits: 1 MFMAs, 2 weight loads, 4 x DMA, 8 epilogues): bash tools/hc_ablate.sh 0 1 2 4 8 15
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for A in "$@"; do
  export PASN_HC_ABL=$A
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hc -o p -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-clips 0 --no-secondary --no-roofline > $R/gpurun_out/prof_hc.log 2>&1 || true
  f=$(find $R/gpurun_out/prof_hc -name "*kernel_stats.csv" | head -1)
  echo "abl=$A $(grep xproto_chain $f | cut -d, -f2-4,6,7)"
  rm -rf $R/gpurun_out/prof_hc
done
