"""Per-launch timing of the split-f16 transposed conv against the fp32 kernel at the HiFi-GAN light upsampler sizes
(tuning aid; knobs: FV_CONVH_BLOCKS, FV_TUNING=1 FV_PAIR_DBG=...).  python tools/convt_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd import _native  # noqa: E402
from pair_bench import bench  # noqa: E402

SHAPES = [(256, 128, 1000, 8, 4, 0), (128, 64, 8000, 5, 3, 1), (512, 256, 1000, 8, 4, 0), (256, 256, 4000, 4, 2, 0)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    for cin, cout, T, s, pad, op in SHAPES:
        k = 2 * s
        x = torch.randn((B, cin, T), generator=g).to(dev)
        w = (torch.randn((cin, cout, k), generator=g) / (2 * cin) ** 0.5).to(dev)
        b = torch.randn(cout, generator=g).to(dev)
        ph = _native.pack_conv_transpose1d_split(w, s)
        pf = _native.pack_conv_transpose1d(w, s, pad)
        tout = (T - 1) * s - 2 * pad + k + op
        y = torch.empty((B, cout, tout), device=dev)
        fl = 2.0 * B * T * cin * cout * k
        ush = bench(lambda: _native.conv_transpose1d_split_f16(x, ph, b, cout, k, s, pad, op, pre_slope=0.1, out=y))
        usf = bench(lambda: _native.conv_transpose1d_fused(x, pf, b, cout, k, s, pad, op, pre_slope=0.1, out=y))
        print(f"convT {cin}->{cout} T={T} s={s} B={B} dbg={os.environ.get('FV_PAIR_DBG', '0')} blocks={os.environ.get('FV_CONVH_BLOCKS', '-')}: "
              f"split {ush:7.1f} us {fl / ush / 1e6:6.1f} TFLOP/s   fp32 {usf:7.1f} us {fl / usf / 1e6:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
