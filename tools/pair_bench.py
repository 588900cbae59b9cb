"""Per-launch timing of the fused ResBlock-pair kernels at the HiFi-GAN light stage sizes
(tuning aid; knobs: FV_PAIR_BLOCKS, FV_PAIR_DBG, FV_PAIRH_SKEL).  python tools/pair_bench.py [C] [T] [B] [f32|split]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd import _native  # noqa: E402
import _ablib  # noqa: E402
_ablib.use_lib_from_env(_native)


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    T = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else {16: 240000, 32: 120000, 64: 40000, 128: 8000}[C]
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    prec = _native.PAIR_SPLIT_F16 if len(sys.argv) > 4 and sys.argv[4] == "split" else _native.PAIR_F32
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    ks = [11, 7, 3]
    xs = [torch.randn((B, C, T), generator=g).to(dev) for _ in ks]
    w1 = [_native.pack_pair((torch.randn((C, C, k), generator=g) / (C * k) ** 0.5).to(dev), prec) for k in ks]
    w2 = [_native.pack_pair((torch.randn((C, C, k), generator=g) / (C * k) ** 0.5).to(dev), prec) for k in ks]
    bs = [torch.randn(C, generator=g).to(dev) for _ in ks]
    ys = [torch.empty_like(x) for x in xs]
    tag = f"{'split' if prec else 'f32'} C={C} T={T} B={B} blocks={os.environ.get('FV_PAIR_BLOCKS', '-')} dbg={os.environ.get('FV_PAIR_DBG', '0')}"
    for dil in (1, 3, 5):
        us = bench(lambda: _native.resblock1_fused(xs, w1, w2, bs, bs, ks, dil, 0.1, outs=ys, prec=prec))
        fl = sum(2 * 2 * B * C * C * k * T for k in ks)
        print(f"{tag} pairs dil={dil}: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
        for j, k in enumerate(ks):
            us = bench(lambda: _native.resblock1_fused([xs[j]], [w1[j]], [w2[j]], [bs[j]], [bs[j]], [k], dil, 0.1,
                                                       outs=[ys[j]], prec=prec))
            print(f"{tag}   member k={k:2d} alone: {us:8.1f} us  {2 * 2 * B * C * C * k * T / us / 1e6:6.1f} TFLOP/s")
    if prec:
        # the end of a stage: the 11- and 7-tap pairs in one launch, then the 3-tap pair with the MRF merge
        out = torch.empty_like(xs[0])
        us2 = bench(lambda: _native.resblock1_fused(xs[:2], w1[:2], w2[:2], bs[:2], bs[:2], ks[:2], 5, 0.1, outs=ys[:2],
                                                    prec=prec))
        us1 = bench(lambda: _native.resblock1_fused([xs[2]], [w1[2]], [w2[2]], [bs[2]], [bs[2]], [3], 5, 0.1, outs=[out],
                                                    prec=prec, add1=[ys[0]], add2=[ys[1]], out_div=3.0, act_slope=0.01))
        print(f"{tag} stage end dil=5: {us2:8.1f} + {us1:8.1f} us")
    elif C == 16:
        us = bench(lambda: _native.mrf_stage(xs, w1, w2, bs, bs, ks, 5, 0.1, out=ys[0]))
        print(f"{tag} mrf_stage dil=5: {us:8.1f} us  {sum(2 * 2 * B * C * C * k * T for k in ks) / us / 1e6:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
