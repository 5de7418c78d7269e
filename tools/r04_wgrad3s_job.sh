# round 4: the shared-image three-tap weight gradient of the three-term mode — tests, then the bench lines (A/B: HIPPIE_WGRAD_SHARED=0)
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_split.py -x -q -m gpu > $O/w3s_tests.log 2>&1; rc=$?; echo "split tests rc $rc"; tail -3 $O/w3s_tests.log
[ $rc -eq 0 ] || exit 1
for sh in 1 0; do
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_WGRAD_SHARED=$sh timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/w3s_b512_$sh.json 2> $O/w3s_b512_${sh}_per_op.txt
  python -c "
import json; d=json.load(open('gpurun_out/r04/w3s_b512_$sh.json')); print('B512 shared=$sh', d['value'], d['ms_per_step'], d['roofline']['wgrad_group_kernel'])"
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_WGRAD_SHARED=$sh timeout -k 10 500 python bench.py --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/w3s_mm_$sh.json 2> $O/w3s_mm_${sh}_per_op.txt
  python -c "
import json; d=json.load(open('gpurun_out/r04/w3s_mm_$sh.json')); print('config5 shared=$sh', d['value'], d['ms_per_step'], d['roofline']['wgrad_group_kernel'])"
done
