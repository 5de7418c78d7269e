# round 4: data-parallel structure — RCCL world-1 tests (Python and plain-C host), then the 1-rank overhead probe for the pair and the multimodal model
set -e
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dp.py tests/test_model_file.py tests/test_gpu_model.py -x -q -m gpu > $O/dp_tests.log 2>&1 || { tail -40 $O/dp_tests.log; exit 1; }
tail -2 $O/dp_tests.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-trainer > $O/dp_uni.json 2> $O/dp_uni.err || { tail -20 $O/dp_uni.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r04/dp_uni.json')); print('uni', d['value'], d['dp_overhead_1rank'])"
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-trainer --no-bucketed-bwd > $O/dp_uni_nb.json 2> $O/dp_uni_nb.err
python -c "
import json; d=json.load(open('gpurun_out/r04/dp_uni_nb.json')); print('uni unbucketed', d['value'], d['dp_overhead_1rank'])"
timeout -k 10 300 python bench.py --model-type multimodal --batch 512 --steps 100 --warmup 10 --no-cpu-baseline --no-trainer > $O/dp_mm512.json 2> $O/dp_mm512.err
python -c "
import json; d=json.load(open('gpurun_out/r04/dp_mm512.json')); print('mm512', d['value'], d['dp_overhead_1rank'])"
timeout -k 10 300 python bench.py --model-type multimodal --batch 512 --steps 100 --warmup 10 --no-cpu-baseline --no-trainer --no-bucketed-bwd > $O/dp_mm512_nb.json 2> $O/dp_mm512_nb.err
python -c "
import json; d=json.load(open('gpurun_out/r04/dp_mm512_nb.json')); print('mm512 unbucketed', d['value'], d['dp_overhead_1rank'])"
