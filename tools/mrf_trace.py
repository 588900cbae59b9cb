"""Phase timeline of the one-launch MRF stage (csrc/mrfh_kernels.hpp; tuning aid: a library built with -DFV_PAIR_TRACE --
tools/build_variant.py trace "-DFV_PAIR_TRACE" -- loaded through FV_AB_LIB).  Per traced block (every 64th), per wave and
tile: shader-clock ticks (s_memtime) between the stamps of mrf_pair -- conv1, epilogue 1, barrier C, conv2, epilogue 2,
vmcnt(0), barrier A -- for the nine pairs of a tile.
    FV_AB_LIB=fastvocoder_amd/libfv_trace.so python tools/mrf_trace.py [B] [fold 0/1] [shape 0/1] [channels 16/32]"""
import os
import sys

import numpy as np
import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fold = int(sys.argv[2]) if len(sys.argv) > 2 else 1
shape = int(sys.argv[3]) if len(sys.argv) > 3 else 0
C = int(sys.argv[4]) if len(sys.argv) > 4 else 16
if C == 32:
    fold = 0
dev = torch.device("cuda:0")
nw = 8 if C == 32 else 16 if shape else 12
trace = torch.zeros(4 * nw * 3 * 64 + 4096, dtype=torch.int64, device=dev)
os.environ["FV_TUNING"] = "1"
os.environ["FV_PAIR_TRACE_PTR"] = hex(trace.data_ptr())
os.environ["FV_MRF_SHAPE"] = str(shape)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fastvocoder_amd import _native  # noqa: E402
import _ablib  # noqa: E402
_ablib.use_lib_from_env(_native)

T, KS = (120000 if C == 32 else 240000), (3, 7, 11)
g = torch.Generator().manual_seed(0)
w1 = [(torch.randn((C, C, KS[q // 3]), generator=g) / (C * KS[q // 3]) ** 0.5).to(dev) for q in range(9)]
w2 = [(torch.randn((C, C, KS[q // 3]), generator=g) / (C * KS[q // 3]) ** 0.5).to(dev) for q in range(9)]
bs = [torch.randn(C, generator=g).to(dev) * 0.1 for _ in range(9)]
P = _native.pack_mrf_stage(w1, w2, bs, bs, list(KS))
x = torch.randn((B, C, T), generator=g).to(dev)
y = torch.empty_like(x)
fw, fb = (torch.randn((16, 7), generator=g) / 10).to(dev), torch.zeros(1, device=dev)
if fold:
    run = lambda: _native.mrf_stage_split_f16(x, P, KS, fold=(fw, fb), act_slope=0.01, post=_native.POST_TANH)  # noqa: E731
else:
    run = lambda: _native.mrf_stage_split_f16(x, P, KS, out=y)  # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
trace.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
print(f"B={B} C={C} fold={fold} shape={shape}: launch (events) {e0.elapsed_time(e1) * 1e3:.1f} us")
tr = trace.cpu().numpy()[:4 * nw * 3 * 64].reshape(4, nw, 3, 64).astype(np.int64)
names = ["conv1", "epi1", "barC", "conv2", "epi2", "vm0", "barA"]
for blk in range(4):
    if tr[blk, 0, 0, 0] == 0:
        continue
    t0 = tr[blk, :, 2, 0].min()
    print(f"block {64 * blk}: entry -> prologue loads landed {int(tr[blk, 0, 2, 1] - tr[blk, 0, 2, 0])} ticks; "
          f"entry -> first pair {int(tr[blk, 0, 0, 0] - t0)}")
    for wave in (0, nw // 2, nw - 1):
        e = tr[blk, wave, 2]
        if e[2]:
            print(f"  tile 0 tail, wave {wave:2d}: sum/div + tile -> LDS {int(e[3] - e[2])}, barrier {int(e[4] - e[3])}, output conv "
                  f"{int(e[5] - e[4])}, barrier {int(e[6] - e[5])}, rest {int(tr[blk, wave, 0, 63] - e[6])}"
                  if e[6] else f"  tile 0 tail, wave {wave:2d}: sum/div + stores {int(tr[blk, wave, 0, 63] - e[2])}")
    for it in range(2):
        if tr[blk, 0, it, 0] == 0:
            continue
        for wave in (0, nw // 2, nw - 1):
            e = tr[blk, wave, it]
            print(f"  tile {it} wave {wave:2d}: total {int(e[63] - e[0])}  (tile end - entry {int(e[63] - t0)})")
            for q in range(9):
                d = [int(e[7 * q + i + 1] - e[7 * q + i]) for i in range(6)] + [int((e[7 * q + 7] if q < 8 else e[63]) - e[7 * q + 6])]
                print(f"     pair {q} (k={KS[q // 3]:2d}, d={(1, 3, 5)[q % 3]}): " + " ".join(f"{n}={v:5d}" for n, v in zip(names, d)) + f"  sum={sum(d)}")
acc, n = np.zeros((9, 7)), 0
for blk in range(4):
    for wave in range(nw):
        for it in range(2):
            e = tr[blk, wave, it]
            if e[0] == 0 or e[63] == 0:
                continue
            for q in range(9):
                acc[q] += [e[7 * q + i + 1] - e[7 * q + i] for i in range(6)] + [(e[7 * q + 7] if q < 8 else e[63]) - e[7 * q + 6]]
            n += 1
if n:
    acc /= n
    print(f"mean over {n} wave-tiles, by phase: " + " ".join(f"{nm}={v:.0f}" for nm, v in zip(names, acc.sum(0))) + f"  tile={acc.sum():.0f}")
    for j in range(3):
        print(f"  ResBlock k={KS[j]}: " + " ".join(f"{nm}={v:.0f}" for nm, v in zip(names, acc[3 * j:3 * j + 3].sum(0))) + f"  sum={acc[3 * j:3 * j + 3].sum():.0f}")
