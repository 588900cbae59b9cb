"""Slot timeline of convq3_kernel (two wave groups one conv phase apart; a library built with -DFV_PAIR_TRACE:
python tools/build_variant.py trace -DFV_PAIR_TRACE, then FV_AB_LIB=fastvocoder_amd/libfv_trace.so): one launch of three
members; traced block 0, every wave: ticks (s_memtime, 100 MHz... shader clock) between the stamps of convq3_run_member.
    python tools/convq3_trace.py T B [k,k,k] [dil] [tuning "key=value,..."]"""
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
import _ablib  # noqa: E402
_ablib.use_lib_from_env(_native)

C = 64
T, B = int(sys.argv[1]), int(sys.argv[2])
ks = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [11, 7, 3]
dil = int(sys.argv[4]) if len(sys.argv) > 4 else 1
for kv in filter(None, (sys.argv[5] if len(sys.argv) > 5 else "").split(",")):
    k_, v_ = kv.split("=")
    _native.tuning_set(k_, int(v_))
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
print(f"C={C} T={T} B={B} ks={ks} dil={dil}: launch (events) {e0.elapsed_time(e1) * 1e3:.1f} us")
tr = trace.cpu().numpy()[:8 * nw * 8 * 16].reshape(8, nw, 8, 16)
names = ["bar1", "K1", "bar2", "E1", "bar3", "K2", "bar4", "vmwait", "E2+st", "convert"]
for blk in (0, 1):
    for w in (0, 4):
        for it in range(2, 5):
            e = tr[blk, w, it]
            if e[0] == 0 or e[10] == 0:
                continue
            d = [int(e[i + 1] - e[i]) for i in range(10)]
            print(f"   block {64 * blk} wave {w} tile {it}: " + " ".join(f"{n}={v}" for n, v in zip(names, d)) + f"  total={int(e[10] - e[0])}")
# the two groups against each other: stamps of wave 0 and wave 4 of block 0 relative to wave 0's tile-1 start
base = tr[0, 0, 1, 0]
for w in (0, 4):
    for it in (1, 2):
        print(f"   wave {w} tile {it} stamps rel: " + " ".join(f"{int(tr[0, w, it, i] - base):6d}" for i in range(11)))
