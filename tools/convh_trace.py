"""Phase timeline of the split-f16 conv kernel (tuning aid; library built with FV_HIPCC_FLAGS=-DFV_PAIR_TRACE):
one launch with FV_PAIR_TRACE_PTR set; per traced block (every 64th), wave 0: cycles between the stamps of
convh_run_member.  python tools/convh_trace.py [C] [k,k,k]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd import _native  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ks = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [11, 7, 3]
T = 40000 if C == 64 else 8000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
S = _native.PAIR_SPLIT_F16
xs = [torch.randn((1, C, T), generator=g).to(dev) for _ in ks]
ws = [_native.pack_pair((torch.randn((C, C, k), generator=g) / (C * k) ** 0.5).to(dev), S) for k in ks]
bs = [torch.randn(C, generator=g).to(dev) for _ in ks]
ys = [torch.empty_like(x) for x in xs]
nw = 8
trace = torch.zeros(8 * nw * 8 * 16 + 1024 * 4, dtype=torch.int64, device=dev)
run = lambda: _native.conv1d_split_f16(xs, ws, bs, ks, 5, pre_slope=0.1, res=xs, outs=ys)  # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
os.environ["FV_PAIR_TRACE_PTR"] = hex(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
print(f"launch (events): {e0.elapsed_time(e1) * 1e3:.1f} us")
del os.environ["FV_PAIR_TRACE_PTR"]
tr = trace.cpu().numpy()[:8 * nw * 8 * 16].reshape(8, nw, 8, 16)
names = ["entry0", "kloop", "bar", "vmwait", "convert", "epi+stores", "to next"]
for blk in range(8):
    ent, stg, ext = tr[blk, 0, 7, 15], tr[blk, 0, 7, 13], tr[blk, 0, 7, 14]
    if ent == 0:
        continue
    t12, t11, t10 = tr[blk, 0, 7, 12], tr[blk, 0, 7, 11], tr[blk, 0, 7, 10]
    print(f"block {64 * blk}: entry->member start {t12 - ent}, loads issued {t11 - t12}, loads landed {t10 - t11}, "
          f"converted {stg - t10}; prologue->exit {ext - stg}, total {ext - ent} ticks")
    for it in range(7):
        e = tr[blk, 0, it]
        if e[0] == 0:
            break
        d = [int(e[i + 1] - e[i]) for i in range(6)]
        nxt = int(tr[blk, 0, it + 1, 0] - e[6]) if tr[blk, 0, it + 1, 0] else 0
        print(f"   tile {it}: " + " ".join(f"{n}={v}" for n, v in zip(names, d + [nxt])) + f"  total={int(e[6] - e[0])}")
