# round 4b: what a K step of the 128-row fragment body costs at config 5's shape — ablated builds (tools/micro/conv_ablate.sh VARIANTS="16 32 256 512 784"; timing only)
# 16 = one product instead of six, 32 = no split arithmetic, 256 = no A staging, 512 = no B fragment loads, 784 = 16 + 256 + 512
O=gpurun_out/r04; mkdir -p $O
for v in 0 16 32 256 512 784; do
  lib=tools/micro/variants/libhippie_abl$v.so; [ $v = 0 ] && lib=hippie_amd/libhippie_hip.so
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_HIP_LIB=$PWD/$lib timeout -k 10 400 python bench.py --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 6 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/babl_$v.json 2> $O/babl_${v}_per_op.txt
  python -c "
import json; d=json.load(open('gpurun_out/r04/babl_$v.json')); r=d['roofline']; print('ablation $v:', round(d['value']), 'samples/s', round(d['ms_per_step'],2), 'ms; conv', round(r['achieved'],1), 'TF')"
  grep -a "pair encoder_mod1.layer4.1.conv1 | encoder_mod2.layer4.1.conv1 " $O/babl_${v}_per_op.txt | head -1 | cut -c1-200
done
