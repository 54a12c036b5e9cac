# R(2+1)D-18[:-3] trunk evidence (GPU box):  bash tools/profile_r2p1d.sh r02b  -> gpurun_out/prof_<tag>_r2p1d/{kernel_stats.csv,bench.json,per_launch_table.txt}
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_${TAG}_r2p1d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --arch resnet2p1d_18 --batch 8 --frames 32 --size 112 --steps 10 --warmup 3 --cpu-clips 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- $CMD > $O/stats.log 2>&1 || echo "stats pass failed"
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cd $R
python3 bench.py --arch resnet2p1d_18 --batch 8 --frames 32 --size 112 --steps 20 --warmup 5 --cpu-clips 0 --per-op $O/per_launch_table.txt > $O/bench.json 2> $O/bench.err
python3 bench.py --arch resnet2p1d_18 --batch 32 --frames 16 --size 224 --steps 5 --warmup 2 --cpu-clips 0 --no-roofline > $O/bench_cfg2_shape.json 2>> $O/bench.err
tail -1 $O/bench.json | cut -c1-300
head -8 $O/kernel_stats.csv
rm -rf $O/stats
