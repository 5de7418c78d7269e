# bench + rocprofv3 kernel trace of the same command (GPU box only).   bash tools/measure_round.sh <tag>
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r02}
cd $R
timeout -k 10 400 python bench.py --steps 300 --warmup 30 --per-op > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_per_op.txt
echo "bench done"; cut -c1-200 gpurun_out/${T}_bench.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o ${T} -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-trainer > $R/gpurun_out/${T}_bench_under_rocprof.json 2> $R/gpurun_out/${T}_rocprof.err
echo "rocprof done"
find $R/gpurun_out/${T}_prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${T}_kernel_stats.csv \;
ls $R/gpurun_out/${T}_prof/* | head
