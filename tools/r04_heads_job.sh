set -e
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_heads.py -x -q -m gpu > $O/heads_tests.log 2>&1 || { tail -40 $O/heads_tests.log; exit 1; }
tail -2 $O/heads_tests.log
timeout -k 10 300 python tools/micro/op_chain_times.py time > $O/op_chain_time_model.txt 2>&1 || { tail -20 $O/op_chain_time_model.txt; exit 1; }
grep -E "HEADS|TOTAL" $O/op_chain_time_model.txt
timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-trainer --no-dp-probe > $O/heads_bench.json 2> $O/heads_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r04/heads_bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('launches_per_pair_step'))"
HIPPIE_DEBUG_KNOBS=1 HIPPIE_NO_FUSE_HEADS=1 timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-trainer --no-dp-probe > $O/heads_bench_off.json 2> $O/heads_bench_off.err
python -c "
import json; d=json.load(open('gpurun_out/r04/heads_bench_off.json')); print('bench unfused', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('launches_per_pair_step'))"
