"""The transposed-conv upsamplers of HiFi-GAN light / MelGAN at small batch: the lean kernel (csrc/convtl_kernels.hpp) against
the ring pipeline (convt_kernel), interleaved rounds, minimum and median per call.   python tools/convt_lean_bench.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd import _native  # noqa: E402

# Cin, Cout, Tin, stride, pad, out_pad   (HiFi-GAN light's first three upsamplers; MelGAN's 512 -> 256 x 8 and 256 -> 128 x 8)
SHAPES = [(256, 128, 1000, 8, 4, 0), (128, 64, 8000, 5, 3, 1), (64, 32, 40000, 3, 2, 1), (512, 256, 200, 8, 4, 0), (256, 128, 1600, 8, 4, 0)]


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
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
    Bs = [int(v) for v in sys.argv[1:]] or [1, 4, 16]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    for B in Bs:
        for cin, cout, T, s, pad, op in SHAPES:
            k = 2 * s
            x = torch.randn((B, cin, T), generator=g).to(dev)
            w = (torch.randn((cin, cout, k), generator=g) / (2 * cin) ** 0.5).to(dev)
            b = torch.randn(cout, generator=g).to(dev)
            ph = _native.pack_conv_transpose1d_split(w, s)
            tout = (T - 1) * s - 2 * pad + k + op
            y = torch.empty((B, cout, tout), device=dev)
            fn = lambda: _native.conv_transpose1d_split_f16(x, ph, b, cout, k, s, pad, op, pre_slope=0.1, out=y)  # noqa: E731
            ts = {"lean": [], "ring": []}
            for _ in range(5):
                for name, v in (("lean", 1 << 20), ("ring", 0)):
                    _native.tuning_set("convt_lean", v)
                    ts[name].append(timed(fn))
            _native.tuning_set("convt_lean", 50)
            mb = 4.0 * B * (cin * T + cout * tout) / 1e6
            print(f"convT {cin:3d}->{cout:3d} x{s} T={T:6d} B={B:2d} ({mb:6.1f} MB): lean {min(ts['lean']):7.1f} / {sorted(ts['lean'])[2]:7.1f} us   "
                  f"ring {min(ts['ring']):7.1f} / {sorted(ts['ring'])[2]:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
