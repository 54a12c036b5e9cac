# SQ / memory counters of the training kernels whose name matches a pattern, from the paired training step (optimisation tool).
#   usage (GPU box): bash tools/pmc_train_kernel.sh grad_pass tag
# Counter passes never share a run with tracing beyond --kernel-trace.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=${1:-grad_pass}
O=$R/gpurun_out/pmct_${2:-x}
mkdir -p $O
run() { # name counters...
  n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/tools/train_bench.py --steps 1 --warmup 1 --loss reference > $O/$n.log 2>&1 || { echo "pass $n failed (see $O/$n.log)"; exit 1; }
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run c SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE
run d TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
python3 - <<PY
import csv,glob,os,collections
O="$O"; PAT="$PAT"
for f in sorted(glob.glob(O+"/*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k in acc:
        if PAT in k:
            print(f.split("/")[-3], k, {c: round(v/cnt[(k,c)]) for c,v in acc[k].items()}, "launches", max(cnt[(k,c)] for c in acc[k]))
PY
rm -rf $O/a $O/b $O/c $O/d
