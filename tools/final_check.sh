set -e
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/final_tests.log 2>&1 || { tail -30 $O/final_tests.log; exit 1; }
tail -2 $O/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py --steps 300 --warmup 30 --per-op > $O/r04_bench.json 2> $O/r04_per_op.txt
python - <<'P'
import json; d=json.load(open("gpurun_out/r04/r04_bench.json")); r=d["roofline"]
print("bench", d["value"], d["ms_per_step"], r["frac"], r["traffic"], r["traffic_source"], r["back_to_back"]["frac"], d.get("trainer_samples_per_s"), (d.get("dp_overhead_1rank") or {}).get("ratio_to_value"), d["cpu_baseline"]["value"])
P
