# which part of the two-bucket data-parallel step costs what (1 rank): variants of parallel.backward_allreduce
O=gpurun_out/r04; mkdir -p $O
for v in "" whole samestream nocoll; do
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_DP_VARIANT=$v timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-trainer > $O/dpv_$v.json 2> $O/dpv_$v.err
  python -c "
import json,sys; d=json.load(open('gpurun_out/r04/dpv_$v.json')); p=d['dp_overhead_1rank']; print('variant [$v]', round(d['value']), p.get('ms_per_step'), p.get('ratio_to_value'), p.get('error'))"
done
