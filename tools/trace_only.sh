set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r03; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $T -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-trainer --no-dp-probe --no-pick-streams --run-ahead 0 > $O/${T}_bench_under_rocprof.json 2> $O/${T}_rocprof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/${T}_kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec python3 $R/tools/timeline.py {} 50 5 \; > $O/${T}_timeline.txt 2>&1 || true
head -8 $O/${T}_timeline.txt
for m in 0 1; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_m$m -o $T -- python3 $R/bench.py --only-model $m --steps 50 --warmup 5 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile --run-ahead 0 > $O/${T}_bench_only_model$m.json 2> /dev/null
  find $O/prof_m$m -name "*kernel_trace.csv" -exec python3 $R/tools/timeline.py {} 50 5 1 \; > $O/${T}_timeline_only_model$m.txt 2>&1 || true
  rm -rf $O/prof_m$m
  head -4 $O/${T}_timeline_only_model$m.txt | tail -2
done
rm -rf $O/prof
