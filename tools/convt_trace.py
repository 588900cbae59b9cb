"""Phase timeline of the split-f16 transposed conv (convt_kernel: convh_run_member's stamps; tuning aid, a library built with
-DFV_PAIR_TRACE through tools/build_variant.py + FV_AB_LIB): one launch; per traced block (every 64th), wave 0: cycles between
the stamps.   python tools/convt_trace.py [Cin Cout stride T B]"""
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

a = [int(v) for v in sys.argv[1:]]
cin, cout, s, T, B = (a + [128, 64, 5, 8000, 1][len(a):])[:5]
k, pad = 2 * s, s // 2 + s % 2
out_pad = s % 2
g = torch.Generator().manual_seed(0)
x = torch.randn((B, cin, T), generator=g).to(dev)
w = (torch.randn((cin, cout, k), generator=g) / (cin * 2) ** 0.5).to(dev)
bias = torch.randn(cout, generator=g).to(dev)
P = _native.pack_conv_transpose1d_split(w, s)
run = lambda: _native.conv_transpose1d_split_f16(x, P, bias, cout, k, s, pad, out_pad, pre_slope=0.1)  # noqa: E731
for _ in range(3):
    y = run()
torch.cuda.synchronize()
trace.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
print(f"ConvTranspose1d {cin} -> {cout} x{s}, T = {T}, B = {B}: launch (events) {e0.elapsed_time(e1) * 1e3:.1f} us, out {tuple(y.shape)}")
tr = trace.cpu().numpy()[:8 * nw * 8 * 16].reshape(8, nw, 8, 16)
names = ["entry", "kloop", "barrier", "vmwait", "merge+epilogue+stores", "convert next"]
for blk in range(8):
    ent, stg, ext = tr[blk, 0, 7, 15], tr[blk, 0, 7, 13], tr[blk, 0, 7, 14]
    t12, t11, t10 = tr[blk, 0, 7, 12], tr[blk, 0, 7, 11], tr[blk, 0, 7, 10]
    if t12 == 0:
        continue
    print(f"block {64 * blk}: member start -> loads issued {t11 - t12}, landed {t10 - t11}, converted {stg - t10}")
    for it in range(7):
        e = tr[blk, 0, it]
        if e[0] == 0:
            break
        d = [int(e[i + 1] - e[i]) for i in range(6)]
        print(f"   item {it}: " + " ".join(f"{n}={v}" for n, v in zip(names, d)) + f"  total={int(e[6] - e[0])}")
    if blk == 0:
        for it in range(2):
            if tr[0, 0, it, 0]:
                print(f"   item {it}, stamps 0..6 of every wave relative to wave 0's stamp 0:")
                for wv in range(nw):
                    print(f"      wave {wv}: " + " ".join(f"{int(tr[0, wv, it, i] - tr[0, 0, it, 0]):7d}" for i in range(7)))
