# round 4b: two knob sweeps on the final kernels at batch 512 — the big-body threshold (tiles per launch) and the grouped weight gradient's blocks per problem
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile > $O/sw_base.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r04/sw_base.json')); print('B512 default', round(d['value']), d['ms_per_step'])"
for t in 224 256 320; do
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_CONV_BIG_MIN_TILES=$t timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile > $O/sw_thr_$t.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r04/sw_thr_$t.json')); print('B512 big-body threshold $t', round(d['value']), d['ms_per_step'])"
done
for b in 32 96 128; do
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_WGRAD_BLOCKS=$b timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile > $O/sw_wgb_$b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r04/sw_wgb_$b.json')); print('B512 wgrad blocks per problem $b', round(d['value']), d['ms_per_step'])"
done
bash tools/c_host/run_pair_bench.sh 300 30 2>&1 | tail -3
