O=gpurun_out/r04; mkdir -p $O
timeout -k 10 500 python bench.py --dtype bf16 --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/mm_bf16.json 2> $O/mm_bf16_per_op.txt
head -12 $O/mm_bf16_per_op.txt
python -c "
import json; d=json.load(open('gpurun_out/r04/mm_bf16.json')); r=d['roofline']; print('mm bf16', d['value'], d['ms_per_step'], r['achieved'], r['frac'])"
