"""Timeline of the TIMED steps inside a rocprofv3 --kernel-trace of `bench.py --steps K --warmup W` (two adamw_kernel
launches per pair-step: the window runs from the end of launch 2W to the end of launch 2(W+K)): kernels in flight, sum of
kernel durations against the wall time, per-kernel shares.   python tools/timeline.py <kernel_trace.csv> K W [models per step: 1 under --only-model]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
K, W = int(sys.argv[2]), int(sys.argv[3])
MODELS = int(sys.argv[4]) if len(sys.argv) > 4 else 2          # adamw_kernel launches per step: 2 (the pair), 1 under --only-model
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
adam = [e for e in ev if "adamw_kernel" in e[2]]
lo, hi = adam[MODELS * W - 1][1], adam[MODELS * (W + K) - 1][1]
ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
tot = hi - lo
print(f"rocprofv3 --kernel-trace of `python3 bench.py --steps {K} --warmup {W} --no-cpu-baseline --no-trainer`" + (" --only-model <m>" if MODELS == 1 else "") +
      f", the {K} timed steps (between the {MODELS * W}th and the {MODELS * (W + K)}th adamw_kernel):")
print(f"window {tot / 1e6:.3f} ms -> {tot / 1e6 / K:.3f} ms/step under tracing; {len(ev) / K:.1f} kernels/step")
pts = sorted([(s, 1) for s, e, n in ev] + [(e, -1) for s, e, n in ev])
busy = collections.Counter()
cur, last = 0, lo
for t, d in pts:
    busy[min(cur, 3)] += t - last
    cur += d
    last = t
for k in range(4):
    print(f"  {k} kernels in flight: {100 * busy[k] / tot:5.1f} % of the wall time")
dur = collections.Counter()
cnt = collections.Counter()
for s, e, n in ev:
    key = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:50]
    dur[key] += e - s
    cnt[key] += 1
tk = sum(dur.values())
print(f"sum of kernel durations {tk / 1e3 / K:.1f} us/step = {tk / tot:.2f} x wall")
for k, v in dur.most_common(30):
    print(f"  {k:52s} {cnt[k] / K:5.1f}/step {v / 1e3 / K:8.1f} us/step {100 * v / tk:5.1f} %  avg {v / cnt[k] / 1e3:6.1f} us")
