"""Summarise rocprofv3 --pmc counter_collection CSVs (FETCH_SIZE / WRITE_SIZE / TCC passes) for one kernel and stamp the
result with what it was measured on (kernel-source digest, shape, git head): bench.py quotes `roofline.traffic` from it
only when the stamp matches the run.   python tools/pmc_summary.py <root> <kernel substring> <out.json> [git head]"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root, kernel, out = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(list)
by_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        by_kernel[r["Counter_Name"]][r["Kernel_Name"].split("(")[0][:80]].append(float(r["Counter_Value"]))
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in agg.items()}
res = {"kernel": kernel, "launches_sampled": {k: len(v) for k, v in agg.items()}, "mean_per_launch": mean}
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE (KB) reads exactly half of a wide coalesced stream on gfx950 -> x2; WRITE_SIZE exact
    res["hbm_bytes_per_launch"] = (2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0
if "TCC_HIT_sum" in mean:
    res["l2_hit_rate"] = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
from bench import csrc_digest
res["meta"] = {"csrc_digest": csrc_digest(), "batch": 512, "z_dim": 10, "wave_len": 50, "time_len": 100,
               "git_head": sys.argv[4] if len(sys.argv) > 4 else None,
               "command": "rocprofv3 --pmc <counter> -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-trainer --no-profile --no-dp-probe --no-pick-streams (one counter set per pass)"}
res["per_kernel_mean"] = {c: {k: sum(v) / len(v) for k, v in sorted(d.items())} for c, d in by_kernel.items()}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "per_kernel_mean"}))
