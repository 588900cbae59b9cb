"""Per-launch duration distribution of one fused-pair launch (is a slow average a slow kernel or a stalled host?):
python tools/launch_jitter.py [C = 128] [T = 8000] [B = 2] [dil = 1] [launches = 2000]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd import _native  # noqa: E402

C, T, B, dil, N = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 128), (2, 8000), (3, 2), (4, 1), (5, 2000)))
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
ks = [11, 7, 3]
S = _native.PAIR_SPLIT_F16
xs = [torch.randn((B, C, T), generator=g).to(dev) for _ in ks]
w1 = [_native.pack_pair((torch.randn((C, C, k), generator=g) / (C * k) ** 0.5).to(dev), S) for k in ks]
w2 = [_native.pack_pair((torch.randn((C, C, k), generator=g) / (C * k) ** 0.5).to(dev), S) for k in ks]
bs = [torch.randn(C, generator=g).to(dev) for _ in ks]
ys = [torch.empty_like(x) for x in xs]
fn = lambda: _native.resblock1_fused(xs, w1, w2, bs, bs, ks, dil, 0.1, outs=ys, prec=S)  # noqa: E731
for _ in range(10):
    fn()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record()
for i in range(N):
    fn()
    ev[i + 1].record()
torch.cuda.synchronize()
d = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N))
print(f"C={C} T={T} B={B} dil={dil}: {N} launches, us: min {d[0]:.1f} median {d[N // 2]:.1f} p99 {d[int(N * 0.99)]:.1f} "
      f"max {d[-1]:.1f}; launches over 3x the median: {sum(v > 3 * d[N // 2] for v in d)}")
