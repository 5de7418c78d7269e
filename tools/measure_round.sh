set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python bench.py --steps 300 --warmup 30 --per-op > gpurun_out/r01b_bench.json 2> gpurun_out/r01b_per_op.txt
echo "bench done"; cut -c1-200 gpurun_out/r01b_bench.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01b_prof -o r01b -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r01b_bench_under_rocprof.json 2> $R/gpurun_out/r01b_rocprof.err
echo "rocprof done"
ls $R/gpurun_out/r01b_prof/* | head
