# A/B of the multimodal tower zip (planner.Builder.zip_towers): batch 512 (the pipeline's default) and BASELINE config 5's per-rank shape
for zt in 1 0 1 0; do
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_NO_ZIP_TOWERS=$((1-zt)) timeout -k 10 300 python bench.py --model-type multimodal --steps 200 --warmup 20 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('zip $zt batch 512 ', round(d['value']), round(d['ms_per_step'],3))"
done
for zt in 1 0 1 0; do
  HIPPIE_DEBUG_KNOBS=1 HIPPIE_NO_ZIP_TOWERS=$((1-zt)) timeout -k 10 300 python bench.py --model-type multimodal --batch 8192 --z-dim 64 --wave-len 256 --time-len 32 --steps 10 --warmup 2 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('zip $zt batch 8192', round(d['value']), round(d['ms_per_step'],3))"
done
timeout -k 10 600 python tools/long_run_vs_oracle.py 300 128 50 1.0 f32 multimodal 2>&1 | tail -2
