"""Time the 16-channel MRF stage of HiFi-GAN light on the GPU: ONE launch (fv_mrf_stage_split_f16, csrc/mrfh_kernels.hpp,
both block shapes, with and without the folded conv_post) against the four pair launches it replaces.
usage: python tools/stage_bench.py [T [B ...]]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from fastvocoder_amd import _native  # noqa: E402

SPLIT = _native.PAIR_SPLIT_F16
DILS = (1, 3, 5)
KS = (3, 7, 11)


def timed(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3      # us


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 240000
    Bs = [int(v) for v in sys.argv[2:]] or [1, 8]
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    w1 = [torch.from_numpy((rng.randn(16, 16, KS[q // 3]) / np.sqrt(16 * KS[q // 3])).astype(np.float32)).to(dev) for q in range(9)]
    w2 = [torch.from_numpy((rng.randn(16, 16, KS[q // 3]) / np.sqrt(16 * KS[q // 3])).astype(np.float32)).to(dev) for q in range(9)]
    b1 = [torch.from_numpy(rng.randn(16).astype(np.float32) * 0.1).to(dev) for _ in range(9)]
    b2 = [torch.from_numpy(rng.randn(16).astype(np.float32) * 0.1).to(dev) for _ in range(9)]
    P = _native.pack_mrf_stage(w1, w2, b1, b2, list(KS))
    P1 = [_native.pack_pair(w, SPLIT) for w in w1]
    P2 = [_native.pack_pair(w, SPLIT) for w in w2]
    fw = torch.from_numpy((rng.randn(16, 7) / 10).astype(np.float32)).to(dev)
    fb = torch.zeros(1, device=dev)
    for B in Bs:
        x = torch.from_numpy(rng.randn(B, 16, T).astype(np.float32)).to(dev)
        y = torch.empty_like(x)
        bufs = [[torch.empty_like(x) for _ in range(2)] for _ in range(3)]
        parts = [torch.empty_like(x) for _ in range(2)]

        def pairs():
            cur = [x, x, x]
            for p in range(2):
                idx = [3 * j + p for j in range(3)]
                outs = [bufs[j][p] for j in range(3)]
                _native.resblock1_fused(cur, [P1[i] for i in idx], [P2[i] for i in idx], [b1[i] for i in idx],
                                        [b2[i] for i in idx], list(KS), DILS[p], 0.1, prec=SPLIT, outs=outs)
                cur = outs
            _native.resblock1_fused(cur[1:], [P1[5], P1[8]], [P2[5], P2[8]], [b1[5], b1[8]], [b2[5], b2[8]], list(KS[1:]),
                                    DILS[2], 0.1, prec=SPLIT, outs=parts)
            _native.resblock1_fused(cur[:1], [P1[2]], [P2[2]], [b1[2]], [b2[2]], [KS[0]], DILS[2], 0.1, prec=SPLIT,
                                    add1=[parts[0]], add2=[parts[1]], out_div=3.0, outs=[y])
        # variants interleaved over several rounds, the minimum of each (the first timings of a process run 5-8 % slower
        # than the later ones -- clocks, caches: a single pass in a fixed order compares the order, not the variants)
        variants = {"four pair launches": (None, None, pairs)}
        for shape in (0, 1):
            for prio in ((0, 1, 3) if shape == 0 else (1,)):
                variants[f"one launch, shape {shape}, prio {prio}"] = (shape, prio, lambda: _native.mrf_stage_split_f16(x, P, KS, out=y))
                variants[f"  ... with conv_post, shape {shape}, prio {prio}"] = (shape, prio, lambda: _native.mrf_stage_split_f16(
                    x, P, KS, fold=(fw, fb), act_slope=0.01, post=_native.POST_TANH))
        best = {k: [] for k in variants}
        for rnd in range(5):
            for name, (shape, prio, fn) in variants.items():
                if shape is not None:
                    _native.tuning_set("mrf_shape", shape)
                    _native.tuning_set("mrf_prio", prio)
                best[name].append(timed(fn, reps=20, warm=3))
        _native.tuning_set("mrf_shape", 0)
        _native.tuning_set("mrf_prio", 1)
        print(f"B={B} T={T}  (us per call: min / median of 5 interleaved rounds)")
        for name, ts in best.items():
            print(f"   {name:48s} {min(ts):8.1f} {sorted(ts)[len(ts) // 2]:8.1f}", flush=True)


if __name__ == "__main__":
    main()
