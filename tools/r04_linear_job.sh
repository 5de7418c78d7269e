# round 4: the Linear family on the matrix cores — unit tests, then the config-5 and config-3 per-op tables
set -e
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "linear or small_leaf or parallel_group" > $O/linear_tests.log 2>&1 || { tail -40 $O/linear_tests.log; exit 1; }
tail -2 $O/linear_tests.log
timeout -k 10 500 python bench.py --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/mm.json 2> $O/mm_per_op.txt
grep -E "LINEAR|samples" $O/mm_per_op.txt | head -60
python -c "
import json; d=json.load(open('gpurun_out/r04/mm.json')); r=d['roofline']; print('mm', d['value'], d['ms_per_step'], r['frac'])"
timeout -k 10 400 python bench.py --batch 4096 --z-dim 32 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/c3.json 2> $O/c3_per_op.txt
grep -E "^LINEAR" $O/c3_per_op.txt
python -c "
import json; d=json.load(open('gpurun_out/r04/c3.json')); r=d['roofline']; print('c3', d['value'], d['ms_per_step'], r['frac'])"
