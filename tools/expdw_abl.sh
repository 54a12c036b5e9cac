# x3d_expdw kernel time under its timing ablations (PASN_EXPDW_ABL bits: 1 expand MFMAs, 2 expand epilogue arithmetic, 4 x DMA, 8 stencil MFMAs, 16 output epilogue + stores)
#   bash tools/expdw_abl.sh 0 1 2 3 8 16 31      (0 = the product instance)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for A in "$@"; do
  if [ "$A" != "0" ]; then export PASN_EXPDW_ABL=$A PASN_EXPDW_S1=0; else unset PASN_EXPDW_ABL PASN_EXPDW_S1; fi  # (the ablation instances exist at stride 2 only)
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_xe -o p -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-clips 0 --no-secondary --no-roofline > $R/gpurun_out/prof_xe.log 2>&1 || true
  f=$(find $R/gpurun_out/prof_xe -name "*kernel_stats.csv" | head -1)
  echo "abl=$A $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'x3d_expdw' in r['Name']: print(r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us avg  min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1), end='   ')
")"
  rm -rf $R/gpurun_out/prof_xe
done
