# SQ counters of the kernels whose name matches a pattern, taken from the headline bench (optimisation tool).
#   usage (GPU box): bash tools/pmc_bench_kernel.sh x3d_expdw tag
# Counter passes never share a run with tracing beyond --kernel-trace.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=${1:-x3d_expdw}
O=$R/gpurun_out/pmcb_${2:-x}
mkdir -p $O
run() { # name counters...
  n=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-clips 0 --no-secondary --no-roofline > $O/$n.log 2>&1 || echo "pass $n failed"
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA
run b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run c SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY
run d GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA
run e SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE
python3 - <<PY
import csv,glob,os,collections
O="$O"; PAT="$PAT"
for f in sorted(glob.glob(O+"/*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k in acc:
        if PAT in k:
            print(f.split("/")[-3], k, {c: round(v/cnt[(k,c)]) for c,v in acc[k].items()})
PY
rm -rf $O/a $O/b $O/c $O/d $O/e
