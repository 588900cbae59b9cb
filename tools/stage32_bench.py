"""Time the 32-channel MRF stage of HiFi-GAN light on the GPU: ONE launch (fv_mrf_stage_split_f16, csrc/mrfw_kernels.hpp)
against the three pair launches it replaces (the merge then rides in the next upsampler).
usage: python tools/stage32_bench.py [T [B ...]]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from fastvocoder_amd import _native  # noqa: E402
from tools.stage_bench import timed  # noqa: E402

SPLIT = _native.PAIR_SPLIT_F16
DILS = (1, 3, 5)
KS = (3, 7, 11)
C = 32


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
    Bs = [int(v) for v in sys.argv[2:]] or [1, 8]
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    w1 = [torch.from_numpy((rng.randn(C, C, KS[q // 3]) / np.sqrt(C * KS[q // 3])).astype(np.float32)).to(dev) for q in range(9)]
    w2 = [torch.from_numpy((rng.randn(C, C, KS[q // 3]) / np.sqrt(C * KS[q // 3])).astype(np.float32)).to(dev) for q in range(9)]
    b1 = [torch.from_numpy(rng.randn(C).astype(np.float32) * 0.1).to(dev) for _ in range(9)]
    b2 = [torch.from_numpy(rng.randn(C).astype(np.float32) * 0.1).to(dev) for _ in range(9)]
    P = _native.pack_mrf_stage(w1, w2, b1, b2, list(KS))
    P1 = [_native.pack_pair(w, SPLIT) for w in w1]
    P2 = [_native.pack_pair(w, SPLIT) for w in w2]
    for B in Bs:
        x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32)).to(dev)
        y = torch.empty_like(x)
        bufs = [[torch.empty_like(x) for _ in range(3)] for _ in range(3)]

        def pairs():
            cur = [x, x, x]
            for p in range(3):
                idx = [3 * j + p for j in range(3)]
                outs = [bufs[j][p] for j in range(3)]
                _native.resblock1_fused(cur, [P1[i] for i in idx], [P2[i] for i in idx], [b1[i] for i in idx],
                                        [b2[i] for i in idx], list(KS), DILS[p], 0.1, prec=SPLIT, outs=outs)
                cur = outs

        variants = {"three pair launches": (None, pairs)}
        for prio in (0, 1):
            variants[f"one launch, prio {prio}"] = (prio, lambda: _native.mrf_stage_split_f16(x, P, KS, out=y))
        best = {k: [] for k in variants}
        for rnd in range(5):
            for name, (prio, fn) in variants.items():
                if prio is not None:
                    _native.tuning_set("mrf_prio", prio)
                best[name].append(timed(fn, reps=20, warm=3))
        _native.tuning_set("mrf_prio", 1)
        print(f"B={B} T={T}  (us per call: min / median of 5 interleaved rounds)")
        for name, ts in best.items():
            print(f"  {name:28s} {min(ts):8.1f} {sorted(ts)[2]:8.1f}")


if __name__ == "__main__":
    main()
