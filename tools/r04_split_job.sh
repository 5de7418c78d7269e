# round 4: fp32 arithmetic on the bf16 matrix cores (HP_CONV_BF16X3) — op tests, the parity suite with the mode as the default, bench lines
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_split.py -x -q -m gpu -s > $O/split_ops.log 2>&1; echo "split op tests rc $?"; grep -a "error vs fp64" $O/split_ops.log; tail -3 $O/split_ops.log
HIPPIE_DEBUG_KNOBS=1 HIPPIE_MFMA_DTYPE=bf16x3 timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_model.py tests/test_gpu_pipeline.py tests/test_gpu_real_data.py tests/test_backbones.py -q -m gpu > $O/split_e2e.log 2>&1; echo "e2e in split mode rc $?"; tail -15 $O/split_e2e.log
for d in f32 bf16x3; do
  timeout -k 10 300 python bench.py --dtype $d --steps 300 --warmup 30 --no-cpu-baseline --no-trainer --no-dp-probe --per-op > $O/split_b512_$d.json 2> $O/split_b512_${d}_per_op.txt
  python -c "
import json; d=json.load(open('gpurun_out/r04/split_b512_$d.json')); r=d['roofline']; print('B512 $d', d['value'], d['ms_per_step'], r['achieved'], r['back_to_back'])"
done
