"""How much do the two model streams actually overlap?  From a rocprofv3 --kernel-trace of the default bench command: per queue,
busy time, gaps between consecutive kernels (the dependent-launch floor), and the fraction of each queue's kernel time during
which the other model's queue also has a kernel in flight — split by kernel family.
python tools/stream_overlap.py <kernel_trace.csv> K W [one_step_listing.txt]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
K, W = int(sys.argv[2]), int(sys.argv[3])
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
adam = [e for e in ev if "adamw_kernel" in e[2]]
lo, hi = adam[2 * W - 1][1], adam[2 * (W + K) - 1][1]
ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
byq = collections.defaultdict(list)
for s, e, n, q in ev:
    byq[q].append((s, e, n))
qs = sorted(byq, key=lambda q: -len(byq[q]))[:2]
print(f"window {(hi - lo) / 1e6:.2f} ms, {K} steps; queues: " + ", ".join(f"{q}: {len(byq[q])} kernels" for q in byq))


def fam(n):
    for key in ("conv_taps", "wgrad_group", "bn_", "linear", "adamw", "stage_batch", "zero", "stem", "tail"):
        if key in n:
            return key
    return "other"


def overlap(a, others):
    """ns of interval a during which some interval of `others` (sorted) is active"""
    s, e = a
    tot = 0
    for os_, oe in others:
        if oe <= s:
            continue
        if os_ >= e:
            break
        tot += min(e, oe) - max(s, os_)
    return tot


for q in qs:
    mine = byq[q]
    other = [(s, e) for oq in qs if oq != q for s, e, _ in byq[oq]]
    other.sort()
    busy = sum(e - s for s, e, _ in mine)
    gaps = [mine[i + 1][0] - mine[i][1] for i in range(len(mine) - 1)]
    gaps = [g for g in gaps if g < 50_000]
    ov = collections.Counter()
    dur = collections.Counter()
    import bisect
    starts = [o[0] for o in other]
    for s, e, n in mine:
        i = max(0, bisect.bisect_left(starts, s) - 2)
        ov[fam(n)] += overlap((s, e), other[i:])
        dur[fam(n)] += e - s
    print(f"queue {q}: busy {busy / 1e3 / K:.0f} us/step of {(hi - lo) / 1e3 / K:.0f}; median gap between consecutive kernels {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, "
          f"sum of gaps {sum(gaps) / 1e3 / K:.0f} us/step")
    for k, v in dur.most_common():
        print(f"   {k:12s} {v / 1e3 / K:8.1f} us/step, {100 * ov[k] / v:5.1f} % of it beside a kernel of the other model")

if len(sys.argv) > 4:
    # one step of both queues as a merged list: start offset (us), queue, duration (us), workgroups, kernel
    full = {(int(r["Start_Timestamp"]), r["Queue_Id"]): r for r in rows}
    a0, a1 = adam[2 * (W + K // 2) - 1][1], adam[2 * (W + K // 2 + 1) - 1][1]
    with open(sys.argv[4], "w") as f:
        for s, e, n, q in ev:
            if a0 <= s < a1:
                r = full[(s, q)]
                wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
                short = n.replace("(anonymous namespace)::", "").split("(")[0].replace("hp::", "").replace("void ", "")[:70]
                f.write(f"{(s - a0) / 1e3:9.2f} {'  ' if q == qs[0] else '          '}q{q} {(e - s) / 1e3:8.2f} us {wg:6d} wg  lds {r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', '?')):>6s} vgpr {r.get('VGPR_Count', '?'):>3s}+{r.get('Accum_VGPR_Count', '?'):<3s} {short}\n")
