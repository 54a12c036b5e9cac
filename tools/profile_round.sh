# Collect the evidence bench.py's roofline cites, on the GPU box:
#   bash tools/profile_round.sh r01d      -> gpurun_out/prof_r01d/{stats,fetch,write}/..., bench json, per-launch table
# Three separate rocprofv3 runs of the SAME command: --kernel-trace --stats, then --pmc FETCH_SIZE, then --pmc WRITE_SIZE
# (counter passes never share a run with tracing beyond --kernel-trace).
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 10 --warmup 3 --cpu-seconds 1 --cpu-clips 2 --no-secondary"  # the headline workload only (the secondary configs re-use its kernels at other shapes and would blur the per-kernel averages)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- $CMD > $O/stats.log 2>&1 || echo "stats pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $CMD > $O/fetch.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $CMD > $O/write.log 2>&1 || echo "write pass failed"
cd $R
python3 tools/pmc_traffic.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $TAG > $O/pmc_traffic.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 bench.py --steps 20 --warmup 5 --per-op $O/per_launch_table.txt > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-400
head -12 $O/kernel_stats.csv
# the heavy raw traces stay on the box
rm -rf $O/stats $O/fetch $O/write
