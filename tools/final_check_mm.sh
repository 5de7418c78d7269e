set -e
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/final_tests.log 2>&1 || { tail -30 $O/final_tests.log; exit 1; }
tail -2 $O/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python bench.py --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/mm.json 2> $O/mm_per_op.txt
python -c "
import json; d=json.load(open('gpurun_out/r03/mm.json')); r=d['roofline']; print('mm', d['value'], d['ms_per_step'], r['frac'], r['encoder_forward']['phase_frac'], r.get('whole_step_tflops_per_gpu'))"
