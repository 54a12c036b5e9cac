# Free-running kernel timeline of the forward bench under two stencil routings: per-step sum of kernel durations and of the gaps between
# consecutive kernels (rocprofv3 --kernel-trace timestamps).  usage (GPU box): bash tools/trace_gaps.sh > gpurun_out/trace_gaps.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in 14 1000; do
  O=$R/gpurun_out/trace_w$v
  rm -rf $O; mkdir -p $O
  PASN_DWMFMA_MAXW=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-clips 0 --no-roofline > $O/log.txt 2>&1
  python3 - $O $v <<'PY'
import csv,glob,sys,collections
O,v=sys.argv[1],sys.argv[2]
f=glob.glob(O+"/**/*kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# last 5 steps = last 5*83 kernels
n=83*5
rows=rows[-n:]
dur=sum(e-s for s,e,_ in rows); span=rows[-1][1]-rows[0][0]
gaps=sum(max(0,rows[i+1][0]-rows[i][1]) for i in range(len(rows)-1))
print(f"maxw {v}: per step: kernel time {dur/5/1e3:.1f} us, gaps {gaps/5/1e3:.1f} us, span {span/5/1e3:.1f} us")
by=collections.defaultdict(float)
for s,e,k in rows: by[k.split('(')[0][:60]]+= (e-s)/5/1e3
for k in sorted(by,key=by.get,reverse=True)[:8]: print(f"    {by[k]:8.1f} us  {k}")
PY
  rm -rf $O
done
