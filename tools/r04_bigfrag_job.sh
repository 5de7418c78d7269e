# round 4b: weight fragments in the 128-row three-term bodies — op tests with the big bodies forced onto the small shapes (bit-identical with / without), then config 5 / config 3 A/B
O=gpurun_out/r04; mkdir -p $O
HIPPIE_DEBUG_KNOBS=1 HIPPIE_CONV_BIG_MIN_TILES=1 timeout -k 10 300 python -m pytest tests/test_gpu_split.py -x -q -m gpu > $O/bigfrag_tests.log 2>&1; rc=$?; echo "split tests, big bodies forced rc $rc"; grep -a "^E " $O/bigfrag_tests.log | head; tail -1 $O/bigfrag_tests.log
[ $rc -eq 0 ] || exit 1
for d in 1 0; do
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_CONV_BIGSHARED=$d timeout -k 10 500 python bench.py --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/bigfrag_mm_$d.json 2> $O/bigfrag_mm_${d}_per_op.txt
  python -c "
import json; d=json.load(open('gpurun_out/r04/bigfrag_mm_$d.json')); r=d['roofline']; print('config5 bigshared=$d', round(d['value']), d['ms_per_step'], r['achieved'])"
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_CONV_BIGSHARED=$d timeout -k 10 400 python bench.py --batch 4096 --z-dim 32 --wave-len 256 --time-len 32 --units 1000000 --steps 20 --warmup 3 --no-cpu-baseline --no-trainer --no-dp-probe > $O/bigfrag_c3_$d.json 2> /dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r04/bigfrag_c3_$d.json')); r=d['roofline']; print('config3 bigshared=$d', round(d['value']), d['ms_per_step'], r['achieved'])"
done
grep -a "N= 512 K= 512" $O/bigfrag_mm_1_per_op.txt | head -3 | cut -c1-200
