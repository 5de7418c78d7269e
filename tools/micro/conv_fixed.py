"""Fixed cost of a CONV_TAPS launch with parts of the kernel disabled (HIPPIE_HIP_LIB = a tools/micro/conv_ablate.sh variant)."""
import os
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools/micro")
from conv_sweep import time_op, time_null, TapMap, P   # noqa: E402

print(os.environ.get("HIPPIE_HIP_LIB", "product library"), flush=True)
print(f"  null chain {time_null():.2f} us", flush=True)
for name, (B, L, N, K) in {"L4 M=2048 N=512 K=512": (512, 4, 512, 512), "L1 M=12800 N=64 K=64": (512, 25, 64, 64)}.items():
    for fl, fn in ((0, "plain"), (P.CONV_STATS, "stats")):
        row = []
        for nt, kk in ((1, 32), (1, K), (3, K)):
            tm = TapMap(B * L, N, kk, L, L, L, 1, 0, [((t % 3) - 1, t % 3) for t in range(nt)])
            row.append((nt * kk // 32, time_op(tm, fl, 3)))
        print(f"  {name} {fn:6s} " + " ".join(f"{s:3d} steps {t:6.2f} us" for s, t in row), flush=True)
