# round 4b: weight fragments (HP_OP_WFRAG + HP_CONV_WFRAG) on / off at the three bench shapes, same box
O=gpurun_out/r04; mkdir -p $O
for nf in 0 1; do
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_NO_WFRAG=$nf timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/wfab_b512_$nf.json 2> $O/wfab_b512_${nf}_per_op.txt
  python -c "
import json; d=json.load(open('gpurun_out/r04/wfab_b512_$nf.json')); r=d['roofline']; print('B512 no_wfrag=$nf', round(d['value']), d['ms_per_step'], 'conv b2b us', r['back_to_back']['avg_launch_us'], 'eager ms', r['eager_serial_ms_per_pair_step'])"
  grep -a "WFRAG" $O/wfab_b512_${nf}_per_op.txt | head -4
done
for nf in 0 1; do
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_NO_WFRAG=$nf timeout -k 10 400 python bench.py --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe > $O/wfab_mm_$nf.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r04/wfab_mm_$nf.json')); r=d['roofline']; print('config5 no_wfrag=$nf', round(d['value']), d['ms_per_step'], r['achieved'])"
done
