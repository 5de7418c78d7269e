# round 4: bf16-stored activations — tests, then the bf16 lines (config 5 shape, batch 512) with both storage forms
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu -s > $O/bf16_tests.log 2>&1; echo "bf16 tests rc $?"
grep -E "lowest gradient|passed|failed|Error" $O/bf16_tests.log | tail -14
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k "long_run" > $O/longrun_tests.log 2>&1; echo "long-run rc $?"; tail -2 $O/longrun_tests.log
for f in "" "--bf16-f32-storage"; do
  timeout -k 10 500 python bench.py --dtype bf16 $f --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/mm_bf16$f.json 2> $O/mm_bf16${f}_per_op.txt
  head -7 $O/mm_bf16${f}_per_op.txt | tail -6
  python -c "
import json; d=json.load(open('gpurun_out/r04/mm_bf16$f.json')); r=d['roofline']; print('mm bf16 [$f]', d['value'], d['ms_per_step'], r['achieved'], r['frac'])"
  timeout -k 10 300 python bench.py --dtype bf16 $f --steps 300 --warmup 20 --no-cpu-baseline --no-trainer --no-dp-probe > $O/b512_bf16$f.json 2> $O/b512_bf16$f.err
  python -c "
import json; d=json.load(open('gpurun_out/r04/b512_bf16$f.json')); r=d['roofline']; print('B512 bf16 [$f]', d['value'], d['ms_per_step'], r['achieved'])"
done
