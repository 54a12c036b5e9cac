# SQ counter passes over one tools/kbench.py launch:  bash tools/pmc_kbench.sh <grep-pattern> <kbench args...>
# (separate --pmc passes, kernel-trace only -- see the rocprofv3 rules in the round instructions)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=$1; shift
O=$R/gpurun_out/pmck
rm -rf $O; mkdir -p $O
run() { n=$1; shift
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/tools/kbench.py $KARGS > $O/$n.log 2>&1 || echo "pass $n failed"
}
KARGS="$*"
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
run b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run c SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT
run d GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR
run e SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE
run f FETCH_SIZE
run g WRITE_SIZE
run h TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
run i TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum
python3 - "$PAT" <<'PY'
import csv,glob,os,collections,sys
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/pmck"
for f in sorted(glob.glob(O+"/*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k in acc:
        if sys.argv[1] in k:
            print(k[:40], {c: round(v/cnt[(k,c)]) for c,v in acc[k].items()})
PY
