"""Phase timeline of the fused 64-channel pair kernel (tuning aid; library built with FV_HIPCC_FLAGS=-DFV_PAIR_TRACE):
one launch of three members; per traced block (every 64th), wave 0: ticks (s_memtime) between the stamps of
convp_run_member.   python tools/convp_trace.py [k,k,k] [dil]"""
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

ks = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [11, 7, 3]
dil = int(sys.argv[2]) if len(sys.argv) > 2 else 1
C, T = 64, 40000
g = torch.Generator().manual_seed(0)
S = _native.PAIR_SPLIT_F16
xs = [torch.randn((1, C, T), generator=g).to(dev) for _ in ks]
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
print(f"launch (events): {e0.elapsed_time(e1) * 1e3:.1f} us")
tr = trace.cpu().numpy()[:8 * nw * 8 * 16].reshape(8, nw, 8, 16)
names = ["conv1", "mid+bar", "conv2", "bar+vmwait", "epi+stores", "convert"]
for blk in range(8):
    if tr[blk, 0, 0, 0] == 0:
        continue
    t12, t10, t13 = tr[blk, 0, 7, 12], tr[blk, 0, 7, 10], tr[blk, 0, 7, 13]
    print(f"block {64 * blk}: (last run) start -> loads landed {t10 - t12}, converted {t13 - t10}")
    for it in range(7):
        e = tr[blk, 0, it]
        if e[0] == 0:
            break
        d = [int(e[i + 1] - e[i]) for i in range(6)]
        print(f"   tile {it}: " + " ".join(f"{n}={v}" for n, v in zip(names, d)) + f"  total={int(e[6] - e[0])}")
