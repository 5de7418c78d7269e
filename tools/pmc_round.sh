# rocprofv3 PMC passes over the bench command (one counter set per run, no trace domains), GPU box only.
#   bash tools/pmc_round.sh [git head]  -> gpurun_out/pmc_<COUNTER>/... + gpurun_out/r02_conv_pmc.json
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 280 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-trainer --no-profile > $R/gpurun_out/pmc_$tag.log 2>&1
  echo "pass $tag done: $(ls $R/gpurun_out/pmc_$tag | head -3 | tr '\n' ' ')"
done
cd $R
python3 tools/pmc_summary.py gpurun_out conv_taps gpurun_out/r02_conv_pmc.json $1
