# round 4: which body serves a three-term conv launch — sweep of the big-body threshold (tiles per launch) at the three bench shapes
O=gpurun_out/r04; mkdir -p $O
for t in 192 256 1024; do
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_CONV_BIG_MIN_TILES=$t timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile > $O/thr_b512_$t.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r04/thr_b512_$t.json')); print('B512 thr $t', d['value'], d['ms_per_step'])"
done
for t in 256 1024 2048; do
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_CONV_BIG_MIN_TILES=$t timeout -k 10 400 python bench.py --batch 4096 --z-dim 32 --wave-len 256 --time-len 32 --units 1000000 --steps 20 --warmup 3 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile > $O/thr_c3_$t.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r04/thr_c3_$t.json')); print('config3 thr $t', d['value'], d['ms_per_step'])"
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_CONV_BIG_MIN_TILES=$t timeout -k 10 500 python bench.py --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile > $O/thr_mm_$t.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r04/thr_mm_$t.json')); print('config5 thr $t', d['value'], d['ms_per_step'])"
done
