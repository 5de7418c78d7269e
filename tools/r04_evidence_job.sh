# round 4 evidence: per-test counts of tensors that pass parity only through the 3x branch (-> tests/parity_budget.json), flip-budget lines
set -e
O=gpurun_out/r04; mkdir -p $O
rm -f $O/parity_report.jsonl
HIPPIE_PARITY_REPORT=$O/parity_report.jsonl timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/parity_tests.log 2>&1 || { tail -30 $O/parity_tests.log; exit 1; }
tail -2 $O/parity_tests.log
python - <<'P'
import json
rows=[json.loads(l) for l in open("gpurun_out/r04/parity_report.jsonl")]
print(len(rows), "tests call parity;", sum(r["total"] for r in rows), "tensors;", sum(r["slack"] for r in rows), "through the 3x branch")
for r in rows:
    if r["slack"]: print(r)
P
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -q -s -m gpu -k "test_forward_grads_and_step_vs_oracle or test_full_batch_512" 2>&1 | grep -o "\[[^]]*\] leaky-ReLU inputs.*" > $O/flip_budget.txt
cat $O/flip_budget.txt | head -20
