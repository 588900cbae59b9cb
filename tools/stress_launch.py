"""Per-launch latency distribution of a ResBlock pair (tuning aid): python tools/stress_launch.py [C] [dil] [n]
-- looks for rare long launches (a stall inside a kernel shows up as a fat tail that a mean hides)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd import _native  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dil = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
T = {16: 240000, 32: 120000, 64: 40000, 128: 8000}[C]
S = _native.PAIR_SPLIT_F16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
ks = [11, 7, 3]
xs = [torch.randn((1, C, T), generator=g).to(dev) for _ in ks]
w1 = [_native.pack_pair((torch.randn((C, C, k), generator=g) / (C * k) ** 0.5).to(dev), S) for k in ks]
w2 = [_native.pack_pair((torch.randn((C, C, k), generator=g) / (C * k) ** 0.5).to(dev), S) for k in ks]
bs = [torch.randn(C, generator=g).to(dev) for _ in ks]
ys = [torch.empty_like(x) for x in xs]
mids = [torch.empty_like(x) for x in xs] if C >= 64 else None
run = lambda: _native.resblock1_fused(xs, w1, w2, bs, bs, ks, dil, 0.1, outs=ys, prec=S, mids=mids)  # noqa: E731
for _ in range(5):
    run()
torch.cuda.synchronize()
ref = [y.clone() for y in ys]
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
for a, b in ev:
    a.record()
    run()
    b.record()
torch.cuda.synchronize()
us = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
same = all(torch.equal(r, y) for r, y in zip(ref, ys))
print(f"C={C} dil={dil} n={n}: median {np.median(us):.1f} us, p99 {np.percentile(us, 99):.1f}, max {us.max():.1f}; "
      f"launches over 3x median: {(us > 3 * np.median(us)).sum()}; results stable: {same}")
