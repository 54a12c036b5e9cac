# LDS audit of every kernel of the headline forward (and, with "train", of one training step): LDS-array cycles per LDS instruction,
# bank-conflict and unaligned-stall cycles.  A 16-byte LDS access off its alignment is replayed at 64 cycles and does NOT show as a bank conflict
# (round 5: the first Toeplitz build read 21 cycles per instruction here where 4-6 is normal).
#   usage (GPU box): bash tools/pmc_lds_audit.sh tag [train]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ldsaudit_${1:-x}
mkdir -p $O
if [ "$2" = "train" ]; then CMD="python3 $R/tools/train_bench.py --steps 1 --warmup 1 --loss reference"; elif [ "$2" = "r2p1d" ]; then CMD="python3 $R/bench.py --arch resnet2p1d_18 --batch 8 --frames 32 --size 112 --prototypes 40 --classes 4 --steps 2 --warmup 1 --cpu-clips 0 --no-secondary --no-roofline --no-graph"; else CMD="python3 $R/bench.py --steps 2 --warmup 1 --cpu-clips 0 --no-secondary --no-roofline --no-graph"; fi
run() { n=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- $CMD > $O/$n.log 2>&1 || { echo "pass $n failed (see $O/$n.log)"; exit 1; }
}
run a SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL
python3 - <<PY
import csv,glob,collections
O="$O"
tab=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in sorted(glob.glob(O+"/*/**/*counter_collection.csv",recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:90]
        tab[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
print("%-92s %9s %12s %8s %10s %10s" % ("kernel", "launches", "lds insts", "cyc/inst", "conflict%", "unaligned%"))
rows=[]
for k,v in tab.items():
    n=cnt[(k,"SQ_INSTS_LDS")]; li=v.get("SQ_INSTS_LDS",0); la=v.get("SQ_LDS_IDX_ACTIVE",0)
    if li<=0: continue
    rows.append((la/li, k, n, li/n, 100*v.get("SQ_LDS_BANK_CONFLICT",0)/max(la,1), 100*v.get("SQ_LDS_UNALIGNED_STALL",0)/max(la,1)))
for r in sorted(rows, reverse=True):
    print("%-92s %9d %12.0f %8.1f %10.1f %10.1f" % (r[1], r[2], r[3], r[0], r[4], r[5]))
PY
rm -rf $O/a
