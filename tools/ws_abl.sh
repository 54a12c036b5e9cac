# pwconv_ws kernel time per step under its timing ablations (PASN_WS_ABL bits: 1 MFMAs, 2 output stores, 4 x DMA, 8 input transform, 16 residual DMA, 32 all DMA)
#   bash tools/ws_abl.sh 0 1 2 3 8 16     (0 = the product path; results are wrong when a bit is set)
# Runs on the tuning build (protoasnet_amd.build_extension(variant="tuning"): -DPASN_TUNING -DPASN_WS_ABLATE; the product kernel compiles the flag
# tests out: they cost 0.5 % of the step, log entry 96) and WITHOUT the hipGraph replay: every kernel then runs exactly warm-up + steps = 13 times,
# which is the divisor below (with the graph-enabled bench each kernel ran 17 times and the reported us/step were 31 % high: ADVICE round 3).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export PASN_LIB_PATH=$R/protoasnet_amd/lib/libprotoasnet_amd_tuning.so
cd /tmp && export TMPDIR=/tmp
for A in "$@"; do
  if [ "$A" != "0" ]; then export PASN_WS_ABL=$A; else unset PASN_WS_ABL; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ws -o p -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-clips 0 --no-secondary --no-roofline --no-graph > $R/gpurun_out/prof_ws.log 2>&1 || true
  f=$(find $R/gpurun_out/prof_ws -name "*kernel_stats.csv" | head -1)
  echo "abl=$A $(python3 -c "
import csv
tot=0; se=0; pair=0; n=0
for r in csv.DictReader(open('$f')):
    if 'pwconv_ws' in r['Name']:
        t=float(r['TotalDurationNs'])/1e3; c=int(r['Calls']); n=max(n,c)
        tot+=t
        nm=r['Name']
        if 'Lb1ELb1' in nm or '<14, 1, true' in nm or ', true, true' in nm: se+=t
steps=13
print('ws family %.1f us/step (SE-prologue instances %.1f)' % (tot/steps, se/steps))
")"
  rm -rf $R/gpurun_out/prof_ws
done
