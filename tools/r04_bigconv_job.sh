# round 4: the large-tile bf16 conv bodies — forced onto the small unit-test shapes, then the config-5 bf16 line
set -e
O=gpurun_out/r04; mkdir -p $O
HIPPIE_DEBUG_KNOBS=1 HIPPIE_CONV_BIG_MIN_TILES=1 timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "conv_taps or layouts or wgrad" > $O/bigconv_tests.log 2>&1 || { tail -40 $O/bigconv_tests.log; exit 1; }
tail -2 $O/bigconv_tests.log
timeout -k 10 500 python bench.py --dtype bf16 --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/mm_bf16_big.json 2> $O/mm_bf16_big_per_op.txt
head -6 $O/mm_bf16_big_per_op.txt
python -c "
import json; d=json.load(open('gpurun_out/r04/mm_bf16_big.json')); r=d['roofline']; print('mm bf16 big', d['value'], d['ms_per_step'], r['achieved'], r['frac'])"
