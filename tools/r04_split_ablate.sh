# round 4b: what the three-term K step costs at batch 512 — bench with ablated builds of conv_mfma.hip (tools/micro/conv_ablate.sh, VARIANTS="16 32 64 112";
# timing only, the numbers are wrong): 16 = only the leading product, 32 = no split arithmetic, 64 = one image in LDS, 112 = all three
O=gpurun_out/r04; mkdir -p $O
for v in ${VLIST:-0 16 32 64 112}; do
  lib=tools/micro/variants/libhippie_abl$v.so; [ $v = 0 ] && lib=hippie_amd/libhippie_hip.so
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-trainer --no-dp-probe > $O/abl_$v.json 2> $O/abl_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/r04/abl_$v.json')); r=d['roofline']; print('ablation $v:', round(d['value']), 'samples/s', round(d['ms_per_step'],3), 'ms; conv avg launch (back to back)', round(r['back_to_back']['avg_launch_us'],2), 'us; wgrad group', round(r['wgrad_group_kernel']['us_per_step']), 'us')"
done
