# Instruction-cache counters per kernel instance of one bench.py run (GPU box):  bash tools/pmc_icache.sh [bench args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_icache
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES --output-format csv -d $O/a -o p -- python3 $R/bench.py --steps 3 --warmup 2 --cpu-clips 0 --no-roofline "$@" > $O/a.log 2>&1 || echo "pass a failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_VALU --output-format csv -d $O/b -o p -- python3 $R/bench.py --steps 3 --warmup 2 --cpu-clips 0 --no-roofline "$@" > $O/b.log 2>&1 || echo "pass b failed"
python3 - <<'PY'
import csv,glob,os,collections,sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]+"/tools")
from pmc_traffic import pretty
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_icache"
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in sorted(glob.glob(O+"/*/**/*counter_collection.csv",recursive=True)):
    for r in csv.DictReader(open(f)):
        k=pretty(r["Kernel_Name"])
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
print("%-44s %8s %10s %10s %8s %10s %8s"%("kernel","launches","req/wave","miss/wave","miss%","ifetch/w","wait%"))
for k in sorted(acc, key=lambda k:-acc[k].get("SQ_WAVE_CYCLES",0)):
    a=acc[k]; w=max(a.get("SQ_WAVES",1),1)
    if "kernel" not in k: continue
    print("%-44s %8d %10.1f %10.1f %7.1f%% %10.1f %7.1f%%"%(k[:44],cnt[(k,"SQ_WAVES")],a.get("SQC_ICACHE_REQ",0)/w,a.get("SQC_ICACHE_MISSES",0)/w,
        100*a.get("SQC_ICACHE_MISSES",0)/max(a.get("SQC_ICACHE_REQ",1),1), a.get("SQ_IFETCH",0)/w, 100*a.get("SQ_WAIT_INST_ANY",0)/max(a.get("SQ_WAVE_CYCLES",1),1)))
PY
rm -rf $O/a $O/b
