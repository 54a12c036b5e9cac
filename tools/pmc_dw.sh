# PMC profile of ONE depthwise layer through tools/kbench.py (optimisation tool).  usage (GPU box): bash tools/pmc_dw.sh "54 1 16 56 56" tag
# env PASN_DWM2 etc. pass through.  Counter passes never share a run with tracing beyond --kernel-trace.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CFG=${1:-"108 1 16 28 28"}
O=$R/gpurun_out/pmcdw_${2:-x}
mkdir -p $O
run() { # name counters...
  n=$1; shift
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/tools/kbench.py dw $CFG > $O/$n.log 2>&1 || echo "pass $n failed"
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
run b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run c SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY
run d GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run e TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv,glob,os,collections
O="$O"
for f in sorted(glob.glob(O+"/*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k in acc:
        if "dwconv" in k:
            print(f.split("/")[-3], k, {c: round(v/cnt[(k,c)]) for c,v in acc[k].items()})
PY
rm -rf $O/a $O/b $O/c $O/d $O/e
