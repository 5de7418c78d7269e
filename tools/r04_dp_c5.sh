# 1-rank data-parallel overhead at config 5's per-rank shape, two-bucket form against one collective after the pass
O=gpurun_out/r04; mkdir -p $O
for f in "" "--bucketed-bwd"; do
  timeout -k 10 500 python bench.py --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer $f > $O/dp_c5$f.json 2> $O/dp_c5$f.err
  python -c "
import json; d=json.load(open('gpurun_out/r04/dp_c5$f.json')); p=d['dp_overhead_1rank']; print('c5 [$f]', round(d['value']), d['ms_per_step'], p.get('ms_per_step'), p.get('ratio_to_value'), p.get('error'))"
done
