"""Phase timeline of the fused pairs without the weight ring (tuning aid; a library built with -DFV_PAIR_TRACE:
python tools/build_variant.py trace -DFV_PAIR_TRACE, then FV_AB_LIB=fastvocoder_amd/libfv_trace.so): one launch of three
members; per traced block (every 64th), wave 0: ticks (s_memtime) between the stamps of convq2_run_member, and their sums.
    python tools/convq2_trace.py C T B [k,k,k] [dil] [tuning "key=value,..."]"""
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

C, T, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ks = [int(a) for a in sys.argv[4].split(",")] if len(sys.argv) > 4 else [11, 7, 3]
dil = int(sys.argv[5]) if len(sys.argv) > 5 else 1
for kv in filter(None, (sys.argv[6] if len(sys.argv) > 6 else "").split(",")):
    k, v = kv.split("=")
    _native.tuning_set(k, int(v))
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
names = ["bar+conv1", "mid+bar", "conv2", "bar+carry+vmwait", "epi+stores", "convert"]
tot = [0] * 6
for blk in range(8):
    if blk < 4 and tr[blk, 0, 7, 12]:
        t12, t10, t13 = tr[blk, 0, 7, 12], tr[blk, 0, 7, 10], tr[blk, 0, 7, 13]
        print(f"   block {64 * blk}: (last run) start -> loads landed {t10 - t12}, converted {t13 - t10}, first tile starts {tr[blk, 0, 0, 0] - t13}")
    for it in range(7):
        e = tr[blk, 0, it]
        if e[0] == 0 or e[6] == 0:
            break
        d = [int(e[i + 1] - e[i]) for i in range(6)]
        if blk < 2 and it < 6:
            print(f"   block {64 * blk} tile {it}: " + " ".join(f"{n}={v}" for n, v in zip(names, d)) + f"  total={int(e[6] - e[0])}")
        if it > 0:                                     # (the LAST member's tiles overwrite the earlier members': it restarts at 0)
            tot = [a + b for a, b in zip(tot, d)]
for it in (0, 2):                                      # the eight waves of block 0 against wave 0's tile start
    if tr[0, 0, it, 0] and tr[0, 0, it, 6]:
        print(f"   block 0 tile {it}, stamps 0..6 of every wave relative to wave 0's stamp 0:")
        for w in range(nw):
            print(f"      wave {w}: " + " ".join(f"{int(tr[0, w, it, i] - tr[0, 0, it, 0]):7d}" for i in range(7)))
s = sum(tot) or 1
print("   shares (tiles 1..): " + " ".join(f"{n}={v / s:.3f}" for n, v in zip(names, tot)))
