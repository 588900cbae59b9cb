"""Phase timeline of the split-f16 fused pair kernel at 16 / 32 channels (tuning aid; a library built with -DFV_PAIR_TRACE,
e.g. tools/build_variant.py trace "-DFV_PAIR_TRACE", loaded through FV_AB_LIB): one launch of the members; per traced
block (every 64th), per wave: ticks (s_memtime, 100 MHz) between the stamps of pairh_run_member.
    FV_AB_LIB=fastvocoder_amd/libfv_trace.so python tools/pairh_trace.py [C] [k,k,k] [dil] [B]"""
import os
import sys

import numpy as np
import torch

C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ks = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [11, 7, 3]
dil = int(sys.argv[3]) if len(sys.argv) > 3 else 3
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
nw = 8 if C == 16 else 15
trace = torch.zeros(8 * nw * 8 * 16 + 1024 * 4, dtype=torch.int64, device=dev)
os.environ["FV_TUNING"] = "1"
os.environ["FV_PAIR_TRACE_PTR"] = hex(trace.data_ptr())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fastvocoder_amd import _native  # noqa: E402
import _ablib  # noqa: E402
_ablib.use_lib_from_env(_native)

T = 240000 if C == 16 else 120000
g = torch.Generator().manual_seed(0)
S = _native.PAIR_SPLIT_F16
xs = [torch.randn((B, C, T), generator=g).to(dev) for _ in ks]
w1 = [_native.pack_pair((torch.randn((C, C, k), generator=g) / (C * k) ** 0.5).to(dev), S) for k in ks]
w2 = [_native.pack_pair((torch.randn((C, C, k), generator=g) / (C * k) ** 0.5).to(dev), S) for k in ks]
bs = [torch.randn(C, generator=g).to(dev) for _ in ks]
ys = [torch.empty_like(x) for x in xs]
run = lambda: _native.resblock1_fused(xs, w1, w2, bs, bs, ks, dil, 0.1, outs=ys, prec=S)  # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
trace.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
print(f"C={C} ks={ks} dil={dil} B={B}: launch (events) {e0.elapsed_time(e1) * 1e3:.1f} us   (1 tick = 10 ns)")
tr = trace.cpu().numpy()[:8 * nw * 8 * 16].reshape(8, nw, 8, 16)
names = ["conv1", "epi1", "barC", "conv2", "vm0", "convert", "epi2", "stores", "barA"]
for blk in range(8):
    if tr[blk, 0, 0, 0] == 0:
        continue
    ent, t12, t13 = tr[blk, 0, 7, 15], tr[blk, 0, 7, 12], tr[blk, 0, 7, 13]
    print(f"block {64 * blk}: entry -> member start {t12 - ent}, prologue (weights, first window, conversion) {t13 - t12}")
    for wave in (0, nw - 1):
        for it in range(7):
            e = tr[blk, wave, it]
            if e[0] == 0 or e[8] == 0:
                break
            d = [int(e[i + 1] - e[i]) for i in range(8)] + [int(e[9] - e[8]) if e[9] else 0]
            print(f"   wave {wave:2d} tile {it}: " + " ".join(f"{n}={v}" for n, v in zip(names, d)) + f"  total={int(max(e[9], e[8]) - e[0])}")
# all waves, all tiles: the mean share of each phase
acc, n = np.zeros(9), 0
for blk in range(8):
    for wave in range(nw):
        for it in range(7):
            e = tr[blk, wave, it]
            if e[0] == 0 or e[9] == 0:
                continue
            acc += np.array([e[i + 1] - e[i] for i in range(9)], dtype=np.float64)
            n += 1
if n:
    print("mean ticks per tile over", n, "wave-tiles:", " ".join(f"{nm}={v / n:.0f}" for nm, v in zip(names, acc)), f" total={acc.sum() / n:.0f}")
