"""Timeline analysis of a rocprofv3 --kernel-trace CSV: how much of the wall time has 0 / 1 / 2+ kernels in flight,
per-queue gaps between consecutive kernels, kernel-time shares.   python tools/trace_gaps.py <kernel_trace.csv> [skip_frac]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows]
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t0 + int((t1 - t0) * skip)          # skip warm-up / set-up
ev = [e for e in ev if e[0] >= lo]
# the timed region ends where the per-op profile (eager, serial, one stream) begins: keep the densest part = until the
# last wgrad_group kernel that overlaps another kernel... simpler: use everything after `lo` and report
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, n, q in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = collections.Counter()
cur, last = 0, pts[0][0]
for t, d in pts:
    busy[min(cur, 3)] += t - last
    cur += d; last = t
tot = t1 - t0
print(f"window {tot/1e6:.2f} ms, kernels {len(ev)}")
for k in sorted(busy):
    print(f"  {k}{'+' if k == 3 else ' '} kernels in flight: {100*busy[k]/tot:5.1f} %")
byq = collections.defaultdict(list)
for e in ev:
    byq[e[3]].append(e)
for q, lst in byq.items():
    lst.sort()
    gaps = [b[0] - a[1] for a, b in zip(lst, lst[1:])]
    g = sorted(x for x in gaps if x > 0)
    dur = sum(e[1] - e[0] for e in lst)
    if len(lst) > 50:
        print(f"queue {q}: {len(lst)} kernels, busy {dur/1e6:.2f} ms, positive gaps: n={len(g)} median {g[len(g)//2]/1e3 if g else 0:.2f} us mean {sum(g)/max(1,len(g))/1e3:.2f} us total {sum(g)/1e6:.2f} ms")
share = collections.Counter()
for s, e, n, q in ev:
    key = n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:40]
    share[key] += e - s
tk = sum(share.values())
for k, v in share.most_common(14):
    print(f"  {k:42s} {v/1e6:8.2f} ms {100*v/tk:5.1f} %")
