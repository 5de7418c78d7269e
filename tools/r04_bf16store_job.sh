# round 4: the bf16 mode with bf16-stored activations (default for --dtype bf16) against fp32 storage — tests, then both lines at config 5's shape and batch 512
O=gpurun_out/r04; mkdir -p $O
HIPPIE_DEBUG_KNOBS=1 HIPPIE_CONV_BIG_MIN_TILES=1 timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "conv_taps or layouts or wgrad or stored" > $O/bf16s_forced.log 2>&1; echo "forced-big rc $?"; tail -1 $O/bf16s_forced.log
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu > $O/bf16_tests.log 2>&1; echo "bf16 tests rc $?"; tail -1 $O/bf16_tests.log
for f in "" "--bf16-f32-storage"; do
  timeout -k 10 500 python bench.py --dtype bf16 $f --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/mm_bf16$f.json 2> $O/mm_bf16${f}_per_op.txt
  head -6 $O/mm_bf16${f}_per_op.txt | tail -5
  python -c "
import json; d=json.load(open('gpurun_out/r04/mm_bf16$f.json')); r=d['roofline']; print('mm bf16 [$f]', d['value'], d['ms_per_step'], r['achieved'], d.get('activation_storage'))"
  timeout -k 10 300 python bench.py --dtype bf16 $f --steps 300 --warmup 20 --no-cpu-baseline --no-trainer --no-dp-probe > $O/b512_bf16$f.json 2> $O/b512_bf16$f.err
  python -c "
import json; d=json.load(open('gpurun_out/r04/b512_bf16$f.json')); r=d['roofline']; print('B512 bf16 [$f]', d['value'], d['ms_per_step'], r['achieved'])"
done
timeout -k 10 400 python bench.py --dtype bf16 --batch 4096 --z-dim 32 --wave-len 256 --time-len 32 --units 1000000 --steps 20 --warmup 3 --no-cpu-baseline --no-trainer --no-dp-probe > $O/c3_bf16.json 2> $O/c3_bf16.err
python -c "
import json; d=json.load(open('gpurun_out/r04/c3_bf16.json')); r=d['roofline']; print('config3 bf16', d['value'], d['ms_per_step'], r['achieved'])"
