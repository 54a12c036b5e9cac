# Training-step evidence on the GPU box:  bash tools/profile_train.sh r01e
#   -> gpurun_out/prof_train_<tag>/{kernel_stats.csv, per_launch.txt, bench_bf16.json, bench_f32.json}
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_train_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/tools/train_bench.py --steps 5 --warmup 2 > $O/stats.log 2>&1 || echo "stats pass failed"
cd $R
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
python3 tools/train_bench.py --steps 10 --warmup 3 --dtype bf16 --per-op $O/per_launch.txt > $O/bench_bf16.json 2> $O/bench.err
python3 tools/train_bench.py --steps 5 --warmup 2 --dtype f32 > $O/bench_f32.json 2>> $O/bench.err
tail -1 $O/bench_bf16.json | cut -c1-300
tail -1 $O/bench_f32.json | cut -c1-300
head -16 $O/kernel_stats.csv
