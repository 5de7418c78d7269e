# rocprofv3 PMC passes for matrix-pipe and LDS utilisation of the step's kernels (derived counters MfmaUtil, LdsUtil, LdsBankConflict; one set per run,
# no trace domains), at batch 512 and at BASELINE config 3's shape.   bash tools/pmc_util.sh <tag>   -> gpurun_out/<tag>/<tag>_pmc_util.txt
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r04b}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp
for shape in b512 c3; do
  if [ $shape = b512 ]; then A="--steps 4 --warmup 1"; else A="--batch 4096 --z-dim 32 --wave-len 256 --time-len 32 --units 200000 --steps 2 --warmup 1"; fi
  for c in MfmaUtil "LdsUtil LdsBankConflict"; do
    tag=$(echo $c | tr ' ' '_')
    timeout -k 10 280 rocprofv3 --pmc $c --output-format csv -d $O/util_${shape}_$tag -o pmc -- python3 $R/bench.py $A --no-cpu-baseline --no-trainer --no-profile --no-dp-probe --no-pick-streams > $O/util_${shape}_$tag.log 2>&1
    echo "pass $shape $tag done"
  done
done
cd $R
python3 - $O $T <<'P'
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
out = open(f"{root}/{tag}_pmc_util.txt", "w")
out.write("rocprofv3 --pmc MfmaUtil | LdsUtil LdsBankConflict (one set per run) over `bench.py` at batch 512 (--steps 4) and at BASELINE config 3's shape\n"
          "(--batch 4096 --z-dim 32 --wave-len 256 --time-len 32, --steps 2); per kernel: launches sampled, mean of the derived counter over its launches.\n"
          "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs) x 100; LdsUtil = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE x CUs) x 100;\n"
          "LdsBankConflict = SQ_LDS_BANK_CONFLICT / (GRBM_GUI_ACTIVE x CUs) x 100.\n")
for shape in ("b512", "c3"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{root}/util_{shape}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.write(f"\n== {shape}\n{'kernel':66s} {'launches':>8s} {'MfmaUtil':>9s} {'LdsUtil':>8s} {'LdsBankConflict':>15s}\n")
    for k, d in sorted(agg.items(), key=lambda kv: -len(kv[1].get("MfmaUtil", []))):
        m = lambda n: (sum(d[n]) / len(d[n])) if d.get(n) else float("nan")
        if max(len(v) for v in d.values()) >= 2:
            out.write(f"{k:66s} {max(len(v) for v in d.values()):8d} {m('MfmaUtil'):9.1f} {m('LdsUtil'):8.1f} {m('LdsBankConflict'):15.2f}\n")
out.close()
print(open(f"{root}/{tag}_pmc_util.txt").read()[:3000])
P
rm -rf $O/util_b512_* $O/util_c3_*
