"""Chunk timeline of the two-source 1x1 GEMM on 128-row tiles (tuning aid; library built with
FV_HIPCC_FLAGS=-DFV_PAIR_TRACE): per traced block (every 64th), wave 0: ticks (s_memtime, ~2.36 GHz observed) between the
stamps of convr_run for the block's first chunks.   python tools/convr_trace.py [C] [B] [T]"""
import os
import sys

import torch

dev = torch.device("cuda:0")
nw = 8
trace = torch.zeros(8 * nw * 8 * 16 + 1024 * 4, dtype=torch.int64, device=dev)
os.environ["FV_TUNING"] = "1"
os.environ["FV_PAIR_TRACE_PTR"] = hex(trace.data_ptr())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd import _native  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
T = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
g = torch.Generator().manual_seed(0)
x, x2 = torch.randn((B, C, T), generator=g).to(dev), torch.randn((B, C, T), generator=g).to(dev)
w1 = (torch.randn((C, C, 1), generator=g) / C ** 0.5).to(dev)
w2 = (torch.randn((C, C, 1), generator=g) / C ** 0.5).to(dev)
P = _native.pack_conv1x1_2src_split(w1, w2)
b = torch.randn(C, generator=g).to(dev)
y = torch.empty_like(x)
_native.tuning_set("convg_rows64", 0)
run = lambda: _native.conv1x1_2src_split_f16(x, x2, P, b, pre_slope=0.2, out=y)  # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
trace.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f"launch (events, traced build): {ms * 1e3:.1f} us = {2.0 * B * C * 2 * C * T / ms / 1e9:.0f} TFLOP/s algorithmic")
tr = trace.cpu().numpy()[:8 * nw * 8 * 16].reshape(8, nw, 8, 16)
names = ["entry0", "4 K steps", "barrier", "epilogue", "raw wait", "convert"]
for blk in range(8):
    if tr[blk, 0, 0, 0] == 0:
        continue
    print(f"block {64 * blk}:")
    for it in range(8):
        e = tr[blk, 0, it]
        if e[0] == 0:
            break
        if e[5] == 0:
            e[5] = e[4]
        d = [int(e[i + 1] - e[i]) for i in range(6)]
        print(f"   chunk {it}: " + " ".join(f"{n}={v}" for n, v in zip(names, d)) + f"  total={int(e[6] - e[0])}")
