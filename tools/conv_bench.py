"""Per-layer-class micro-benchmark of the fused conv kernels (GPU box only).

    python tools/conv_bench.py [--batch B] [--iters N] [--model light|large]

Times each distinct conv shape of the HiFi-GAN generator in isolation with
events on the launch stream and prints achieved TFLOP/s (vs the 157.3 TF fp32
MFMA peak) and algorithmic GB/s (vs ~6.3 TB/s achievable HBM).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd import _native  # noqa: E402


def time_op(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--model", default="light")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--only", default="", help="substring filter on '<name> k<k> d<d>', e.g. 'res C32 k11 d1'")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    c0 = 256 if args.model == "light" else 512
    T = args.frames
    B = args.batch
    rows = []
    # conv_pre
    shapes = [("conv_pre", 80, c0, T, 7, 1, None)]
    t = T
    for i, (u, ku) in enumerate(zip([8, 5, 3, 2], [16, 10, 6, 4])):
        cin, cout = c0 >> i, c0 >> (i + 1)
        shapes.append((f"convT s{u}", cin, cout, t, ku, 1, u))
        t *= u
        for k in (3, 7, 11):
            for d in (1, 5):
                shapes.append((f"res C{cout}", cout, cout, t, k, d, None))
    shapes.append(("conv_post", c0 >> 4, 1, t, 7, 1, None))
    tot_t = 0.0
    print(f"{'layer':12s} {'Cin':>4s} {'Cout':>4s} {'T':>7s} {'k':>2s} {'d':>2s} {'us':>8s} {'TF/s':>7s} {'GB/s':>7s}")
    for name, cin, cout, tt, k, d, up in shapes:
        if args.only and args.only not in f"{name} k{k} d{d}":
            continue
        x = torch.randn(B, cin, tt, device=dev)
        if up is None:
            w = torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5
            packed = _native.pack_conv1d(w)
            bias = torch.randn(cout, device=dev)
            pad = (k - 1) * d // 2
            out = torch.empty(B, cout, tt, device=dev)
            res = torch.randn(B, cout, tt, device=dev) if cin == cout else None
            fn = lambda: _native.conv1d_fused(x, packed, bias, cout, k, dil=d, pad=pad, pre_slope=1.0,
                                              res=res, out=out)
            flops = 2.0 * B * cout * tt * cin * k
            byts = 4.0 * B * tt * (cin + cout * (2 if res is not None else 1))
        else:
            w = torch.randn(cin, cout, k, device=dev) / (cin * k / up) ** 0.5
            p = up // 2 + up % 2
            packed = _native.pack_conv_transpose1d(w, up, p)
            bias = torch.randn(cout, device=dev)
            out = torch.empty(B, cout, tt * up, device=dev)
            fn = lambda: _native.conv_transpose1d_fused(x, packed, bias, cout, k, up, p, up % 2,
                                                        pre_slope=1.0, out=out)
            flops = 2.0 * B * tt * cin * cout * k
            byts = 4.0 * B * tt * (cin + cout * up)
        s = time_op(fn, args.iters)
        n = 1
        if name.startswith("res"):
            n = 6 if d == 1 else 3      # per (C,k): 3 convs2 + conv1 d=1 at d=1; d=3,5 ~ d=5 cost
            n = 4.0 if d == 1 else 2.0
        tot_t += s * n
        print(f"{name:12s} {cin:4d} {cout:4d} {tt:7d} {k:2d} {d:2d} {s*1e6:8.1f} {flops/s/1e12:7.1f} {byts/s/1e9:7.0f}")
    print(f"estimated whole-generator kernel time: {tot_t*1e3:.3f} ms")


if __name__ == "__main__":
    main()
