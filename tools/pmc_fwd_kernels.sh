# Issue-side counters of every trunk kernel of the headline forward (optimisation tool): which pipes are busy.
#   usage (GPU box): bash tools/pmc_fwd_kernels.sh tag
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmcf_${1:-x}
mkdir -p $O
run() { n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-clips 0 --no-secondary --no-roofline --no-graph > $O/$n.log 2>&1 || { echo "pass $n failed (see $O/$n.log)"; exit 1; }
}
run a SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES
python3 - <<PY
import csv,glob,collections
O="$O"
tab=collections.defaultdict(dict)
for f in sorted(glob.glob(O+"/*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:64]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k in acc:
        for c,v in acc[k].items(): tab[k][c]=v/cnt[(k,c)]
print("%-66s %8s %6s %6s %6s %6s %6s" % ("kernel", "busy", "valu", "lds", "mfma", "salu", "vmem"))
for k,v in sorted(tab.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES",0)):
    b=v.get("SQ_BUSY_CYCLES",1)
    if "pasn" not in k and "se_gate" not in k: continue
    print("%-66s %8d %6.2f %6.2f %6.2f %6.2f %6.2f" % (k, b, v.get("SQ_ACTIVE_INST_VALU",0)/b, v.get("SQ_ACTIVE_INST_LDS",0)/b, v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/b, v.get("SQ_ACTIVE_INST_SCA",0)/b, v.get("SQ_INST_CYCLES_VMEM",0)/b))
PY
rm -rf $O/a $O/b
