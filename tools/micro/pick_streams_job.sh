set -e
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/pick.json 2> $O/pick.err
python - <<'P'
import json; d=json.load(open("gpurun_out/r03b/pick.json")); print("pick", d["value"], d["ms_per_step"], d["config"]["stream_pair"], d.get("trainer_samples_per_s"), d.get("dp_overhead_1rank"))
P
timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-pick-streams --no-trainer --no-dp-probe > $O/nopick.json 2> $O/nopick.err
python - <<'P'
import json; d=json.load(open("gpurun_out/r03b/nopick.json")); print("nopick", d["value"], d["ms_per_step"], d.get("dp_overhead_1rank"))
P
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -5
