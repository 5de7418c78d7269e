# round 4: HP_CONV_BF16X3 in the 128-row bodies — op tests with the big bodies forced onto the small shapes, then the large-batch lines
O=gpurun_out/r04; mkdir -p $O
HIPPIE_DEBUG_KNOBS=1 HIPPIE_CONV_BIG_MIN_TILES=1 timeout -k 10 600 python -m pytest tests/test_gpu_split.py -x -q -m gpu -s > $O/split_ops_big.log 2>&1; echo "split op tests, big bodies forced rc $?"; grep -a "error vs fp64" $O/split_ops_big.log; tail -3 $O/split_ops_big.log
for d in f32 bf16x3; do
  timeout -k 10 400 python bench.py --dtype $d --batch 4096 --z-dim 32 --wave-len 256 --time-len 32 --units 1000000 --steps 20 --warmup 3 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/split_c3_$d.json 2> $O/split_c3_${d}_per_op.txt
  python -c "
import json; d=json.load(open('gpurun_out/r04/split_c3_$d.json')); r=d['roofline']; print('config3 $d', d['value'], d['ms_per_step'], r['achieved'])"
  timeout -k 10 500 python bench.py --dtype $d --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/split_mm_$d.json 2> $O/split_mm_${d}_per_op.txt
  python -c "
import json; d=json.load(open('gpurun_out/r04/split_mm_$d.json')); r=d['roofline']; print('config5 $d', d['value'], d['ms_per_step'], r['achieved'])"
done
