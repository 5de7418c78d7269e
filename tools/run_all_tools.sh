# bitrot check: every measurement tool still runs on the GPU box (exit status only).  bash tools/run_all_tools.sh
export TMPDIR=/tmp
for f in tools/infer_rate.py tools/numerics_probe.py tools/op_table.py tools/single_model_time.py \
         tools/micro/bn_phases.py tools/micro/bn_sweep.py tools/micro/conv_fixed.py tools/micro/conv_inbn_sweep.py tools/micro/conv_phases.py tools/micro/conv_sweep.py \
         tools/micro/linear_sweep.py tools/micro/small_wgrad_sweep.py tools/micro/wgrad_sweep.py "tools/micro/op_chain_times.py time"; do
  if timeout -k 10 240 python $f > /tmp/tool.out 2>&1; then echo "OK   $f"; else echo "FAIL $f: $(grep -v amdgpu.ids /tmp/tool.out | tail -2 | tr '\n' ' ' | cut -c1-300)"; fi
done
