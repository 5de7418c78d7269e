for v in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --run-ahead 0" "--steps 20 --warmup 5" "--steps 20 --warmup 5 --run-ahead 0" "--steps 300 --warmup 30" "--steps 300 --warmup 30 --run-ahead 0" "--steps 300 --warmup 30"; do
  timeout -k 10 300 python bench.py $v --no-cpu-baseline --no-trainer --no-dp-probe --no-profile 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', round(d['value']), round(d['ms_per_step'],3))"
done
timeout -k 10 600 python -m pytest tests/test_gpu_dp.py -x -q -m gpu -k "bench or rccl" 2>&1 | tail -2
