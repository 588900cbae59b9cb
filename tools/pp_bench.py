"""A/B of the fused 64-channel pair launch: convq2_kernel (convp_pp = 0) against convq3_kernel (two wave groups one conv
phase apart, convp_pp = 1) at HiFi-GAN light's stage size, per three-member launch in a hot loop, and whole forwards.
python tools/pp_bench.py [batch = 1]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fastvocoder_amd import _native  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
C, T = 64, 40000
SPLIT = _native.PAIR_SPLIT_F16


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


ks = (11, 7, 3)
xs = [t(rng.randn(B, C, T) * 0.3) for _ in ks]
w1 = [_native.pack_pair(t(rng.randn(C, C, k) / np.sqrt(C * k)), SPLIT) for k in ks]
w2 = [_native.pack_pair(t(rng.randn(C, C, k) / np.sqrt(C * k)), SPLIT) for k in ks]
b1 = [t(rng.randn(C) * 0.1) for _ in ks]
b2 = [t(rng.randn(C) * 0.1) for _ in ks]
outs = [torch.empty_like(x) for x in xs]
res = {}
for dil in (1, 3, 5):
    for pp in (0, 1, 2):
        _native.tuning_set("convp_pp", pp)
        for _ in range(5):
            ys = _native.resblock1_fused(xs, w1, w2, b1, b2, list(ks), dil, 0.1, prec=SPLIT, outs=outs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            _native.resblock1_fused(xs, w1, w2, b1, b2, list(ks), dil, 0.1, prec=SPLIT, outs=outs)
        torch.cuda.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / n
        res[(dil, pp)] = [y.clone() for y in ys]
        print(f"B={B} dil={dil} pp={pp}: {us:8.1f} us per three-member launch")
    same = all(torch.equal(a, b) and torch.equal(a, c) for a, b, c in zip(res[(dil, 0)], res[(dil, 1)], res[(dil, 2)]))
    print(f"   identical bits: {same}")
bench.T_FRAMES = 1000
model, cfg, sd = bench.build_model("light", dev, None, 0)
mel = torch.from_numpy(bench.utterance_mels(0, B)).to(dev)
for pp in (0, 1, 2, 0, 1, 2):
    _native.tuning_set("convp_pp", pp)
    with torch.no_grad():
        for _ in range(5):
            y = model(mel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            y = model(mel)
        torch.cuda.synchronize()
    print(f"forward B={B} pp={pp}: {1e3 * (time.perf_counter() - t0) / 50:.4f} ms")
