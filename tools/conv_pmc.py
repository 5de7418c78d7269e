"""One conv shape, a few launches: target for rocprofv3 --pmc passes (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.conv_micro import bench
M, N, K, L = (int(v) for v in sys.argv[1:5])
us, tf = bench(M, N, K, L, False, 0, reps=6)
print(f"M={M} N={N} K={K}: {us:.1f} us {tf:.1f} TF")
