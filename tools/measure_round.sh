# One round's measurements on the GPU box: bench + per-op table, rocprofv3 kernel trace of the same command (+ timeline),
# PMC passes (one counter set per run, no trace domains), the labelled secondary lines.   bash tools/measure_round.sh <tag> <git head>
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r04b}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --steps 300 --warmup 30 --per-op > $O/${T}_bench.json 2> $O/${T}_per_op.txt
echo "bench done: $(cut -c1-160 $O/${T}_bench.json)"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $T -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile --no-pick-streams --run-ahead 0 > $O/${T}_bench_under_rocprof.json 2> $O/${T}_rocprof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/${T}_kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec python3 $R/tools/timeline.py {} 50 5 \; > $O/${T}_timeline.txt 2>&1 || true
echo "rocprof done: $(head -3 $O/${T}_timeline.txt | tail -1)"
# (--no-profile in the traced run: the per-launch HIP-event pass and the 20x back-to-back graphs would be counted into rocprof's per-kernel
#  averages — round 3's kernel_stats.csv held 296 conv calls per step instead of 128)
# (--run-ahead 0 in the traced runs: under the tracer a graph launch costs the host milliseconds, and a host that also waits for step i - 2 before
# launching step i leaves the GPU idle 6 % of the time — an artefact of tracing, not of the schedule)
# (--no-pick-streams in the profiler runs: the stream-pair measurement replays the models' EVAL-forward graphs, whose conv launches — same kernel
# names, different epilogue, partly on badly overlapping stream pairs — would be counted into rocprof's per-kernel averages and PMC means)
# the same trace with ONE model stepping alone (its kernels never share the GPU: durations are the kernel's own)
for m in 0 1; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_m$m -o $T -- python3 $R/bench.py --only-model $m --steps 50 --warmup 5 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile --run-ahead 0 > $O/${T}_bench_only_model$m.json 2> /dev/null
  find $O/prof_m$m -name "*kernel_trace.csv" -exec python3 $R/tools/timeline.py {} 50 5 1 \; > $O/${T}_timeline_only_model$m.txt 2>&1 || true
  rm -rf $O/prof_m$m
done
echo "single-model traces done"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 280 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -o pmc -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-trainer --no-profile --no-dp-probe --no-pick-streams > $O/pmc_$tag.log 2>&1
  echo "pmc pass $tag done"
done
cd $R
python3 tools/pmc_summary.py $O conv_taps $O/${T}_conv_pmc.json $2
rm -rf $O/prof $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum_TCC_MISS_sum
timeout -k 10 200 python bench.py --dtype bf16 --steps 200 --warmup 20 --no-cpu-baseline --no-trainer --no-dp-probe > $O/${T}_bench_bf16.json 2> $O/${T}_bench_bf16.err
timeout -k 10 300 python bench.py --batch 4096 --z-dim 32 --wave-len 256 --time-len 32 --units 1000000 --steps 20 --warmup 3 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/${T}_bench_config3_B4096.json 2> $O/${T}_per_op_config3_B4096.txt
timeout -k 10 400 python bench.py --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/${T}_bench_multimodal_B8192.json 2> $O/${T}_multimodal_B8192_per_op.txt
# the same three lines on the fp32 matrix cores (rounds 1-3's kernels): the A/B behind DESIGN section 5.1's "three-term path" column
timeout -k 10 300 python bench.py --matrix-path f32 --steps 300 --warmup 30 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/${T}_bench_f32_matrix.json 2> $O/${T}_per_op_f32_matrix.txt
timeout -k 10 300 python bench.py --matrix-path f32 --batch 4096 --z-dim 32 --wave-len 256 --time-len 32 --units 1000000 --steps 20 --warmup 3 --no-cpu-baseline --no-trainer --no-dp-probe > $O/${T}_bench_config3_B4096_f32_matrix.json 2> /dev/null
timeout -k 10 400 python bench.py --matrix-path f32 --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe > $O/${T}_bench_multimodal_B8192_f32_matrix.json 2> /dev/null
# the reduced-precision mode at config 5's shape and the multimodal model at the pipeline's batch 512
timeout -k 10 400 python bench.py --dtype bf16 --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/${T}_bench_multimodal_B8192_bf16.json 2> $O/${T}_multimodal_B8192_bf16_per_op.txt
timeout -k 10 300 python bench.py --model-type multimodal --batch 512 --steps 200 --warmup 20 --no-cpu-baseline --no-trainer > $O/${T}_bench_multimodal_B512.json 2> /dev/null
timeout -k 10 300 python tools/micro/op_chain_times.py time > $O/${T}_op_chain_time_model.txt 2>&1 || true
timeout -k 10 300 python tools/micro/op_chain_times.py wave > $O/${T}_op_chain_wave_model.txt 2>&1 || true
ls $O
