# round 4: the whole GPU suite with "bf16x3" as the default matrix path, the split op tests with the big bodies forced, smoke, default bench
O=gpurun_out/r04; mkdir -p $O
HIPPIE_DEBUG_KNOBS=1 HIPPIE_CONV_BIG_MIN_TILES=1 timeout -k 10 600 python -m pytest tests/test_gpu_split.py -q -m gpu -s > $O/split_ops_big.log 2>&1; echo "split op tests, big bodies forced rc $?"; grep -a "error vs fp64" $O/split_ops_big.log; tail -3 $O/split_ops_big.log
timeout -k 10 1100 python -m pytest tests -q -m gpu -s > $O/full_tests.log 2>&1; echo "full suite rc $?"; grep -a "outputs .* scalars\|error vs fp64" $O/full_tests.log; tail -12 $O/full_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py --steps 300 --warmup 30 > $O/bench_default.json 2> $O/bench_default.err
python - <<'P'
import json; d=json.load(open("gpurun_out/r04/bench_default.json")); r=d["roofline"]
print("bench", d["value"], d["ms_per_step"], d["matrix_path"], r["frac"], r["frac_of_f32_mfma_peak"], r["back_to_back"].get("frac"), d.get("trainer_samples_per_s"), (d.get("dp_overhead_1rank") or {}).get("ratio_to_value"), d["cpu_baseline"]["value"])
P
