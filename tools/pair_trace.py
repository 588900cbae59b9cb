"""Phase timeline of the fused pair kernel (tuning aid): runs one launch with FV_PAIR_TRACE_PTR set and
prints, per block/wave/tile, the cycles spent between the stamped events (pair_stamp in pair_kernels.hpp).
python tools/pair_trace.py [C] [k,k,k]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd import _native  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ks = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [11, 7, 3]
T = 240000 if C == 16 else 120000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
xs = [torch.randn((1, C, T), generator=g).to(dev) for _ in ks]
w1 = [_native.pack_pair((torch.randn((C, C, k), generator=g) / (C * k) ** 0.5).to(dev)) for k in ks]
w2 = [_native.pack_pair((torch.randn((C, C, k), generator=g) / (C * k) ** 0.5).to(dev)) for k in ks]
bs = [torch.randn(C, generator=g).to(dev) for _ in ks]
ys = [torch.empty_like(x) for x in xs]
nw = 8 if C == 16 else 16
trace = torch.zeros(8 * nw * 8 * 16 + 1024 * 4, dtype=torch.int64, device=dev)
for _ in range(3):
    _native.resblock1_fused(xs, w1, w2, bs, bs, ks, 5, 0.1, outs=ys)
torch.cuda.synchronize()
os.environ["FV_PAIR_TRACE_PTR"] = hex(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_native.resblock1_fused(xs, w1, w2, bs, bs, ks, 5, 0.1, outs=ys)
e1.record()
torch.cuda.synchronize()
print(f"launch (events): {e0.elapsed_time(e1) * 1e3:.1f} us")
del os.environ["FV_PAIR_TRACE_PTR"]
full = trace.cpu().numpy()
tr = full[:8 * nw * 8 * 16].reshape(8, nw, 8, 16)
blk = full[8 * nw * 8 * 16:].reshape(1024, 4)
import collections
groups = collections.defaultdict(list)
for i in range(1024):
    if blk[i, 0] == 0:
        continue
    hw, xcc = int(blk[i, 2]), int(blk[i, 3]) & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    groups[(xcc, se, sh, cu)].append((int(blk[i, 0]), int(blk[i, 1]), i, (hw >> 16) & 0xF))
print(f"{sum(len(v) for v in groups.values())} blocks on {len(groups)} distinct (xcc, se, sh, cu)")
shown = 0
for key, v in sorted(groups.items()):
    v.sort()
    t0 = v[0][0]
    if shown < 6:
        print(key, [(b, tg, s - t0, e - t0) for s, e, b, tg in v])
    shown += 1

names = ["top>A", "act", "act>B", "conv1", "conv1>C", "drain", "issue", "conv2", "vm0", "stores", "next"]
t00 = tr[tr > 0].min()
for blk in range(8):
    for wave in (0,):
        ent, stg, ext = tr[blk, wave, 7, 15], tr[blk, wave, 7, 13], tr[blk, wave, 7, 14]
        print(f"block {64 * blk} wave {wave}: entry->staged {stg - ent} staged->exit {ext - stg} entry->exit {ext - ent} ticks")
        for it in range(8):
            e = tr[blk, wave, it]
            if e[0] == 0:
                break
            d = [int(e[i + 1] - e[i]) for i in range(10)]
            nxt = int(tr[blk, wave, it + 1, 0] - e[10]) if it < 7 and tr[blk, wave, it + 1, 0] else 0
            print(f"   tile {it}: " + " ".join(f"{n}={v}" for n, v in zip(names, d + [nxt])) + f"  total={int(e[10] - e[0])}")
