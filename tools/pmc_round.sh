# rocprofv3 PMC passes over the bench command (one counter set per run, no trace domains), GPU box only.
#   bash tools/pmc_round.sh <tag> [git head]  -> gpurun_out/<tag>/<tag>_conv_pmc.json (what bench.py's roofline.traffic quotes when its digest matches)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r04b}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 280 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -o pmc -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-trainer --no-profile --no-dp-probe --no-pick-streams > $O/pmc_$tag.log 2>&1
  echo "pmc pass $tag done"
done
cd $R
python3 tools/pmc_summary.py $O conv_taps $O/${T}_conv_pmc.json $2
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum_TCC_MISS_sum
