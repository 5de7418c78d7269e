O=gpurun_out/r04; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_split.py -x -q -m gpu -k "fragment" > $O/wfrag_tests.log 2>&1; rc=$?; echo "wfrag op tests rc $rc"; grep -a "^E " $O/wfrag_tests.log | head -20; tail -1 $O/wfrag_tests.log
[ $rc -eq 0 ] || exit 1
for nf in 0 1; do
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_NO_WFRAG=$nf timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-trainer --no-dp-probe > $O/wfab_b512_$nf.json 2> /dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r04/wfab_b512_$nf.json')); r=d['roofline']; print('B512 no_wfrag=$nf', round(d['value']), d['ms_per_step'], 'conv b2b us', r['back_to_back']['avg_launch_us'], 'eager ms', r['eager_serial_ms_per_pair_step'])"
done
