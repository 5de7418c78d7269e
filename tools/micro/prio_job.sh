set -e
O=gpurun_out/r03b; mkdir -p $O
for v in "" "--no-stream-priority" "--no-pick-streams" "" "--no-stream-priority"; do
  timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile $v > $O/prio.json 2>/dev/null
  python - "$v" <<'P'
import json,sys; d=json.load(open("gpurun_out/r03b/prio.json")); print(f"{sys.argv[1] or 'default (priority)':24s} {d['value']:9.0f} samples/s {d['ms_per_step']:.3f} ms")
P
done
timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/prio_full.json 2> $O/prio_full.err
python - <<'P'
import json; d=json.load(open("gpurun_out/r03b/prio_full.json")); print("full", d["value"], d["ms_per_step"], d["config"]["stream_pair"], d.get("trainer_samples_per_s"), (d.get("dp_overhead_1rank") or {}).get("ratio_to_value"))
P
timeout -k 10 300 python -m pytest tests/test_gpu_streams.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -3
