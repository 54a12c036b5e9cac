# rocprofv3 kernel stats of the headline bench, filtered:  bash tools/prof_kernels.sh <tag> <grep -E pattern>
TAG=${1:-x}; PAT=${2:-xproto}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-clips 0 --no-secondary --no-roofline > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R && cp $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv && rm -rf gpurun_out/prof_$TAG
grep -E "$PAT" gpurun_out/${TAG}_kernel_stats.csv | cut -c1-220
tail -1 gpurun_out/prof_$TAG.log | cut -c1-120
