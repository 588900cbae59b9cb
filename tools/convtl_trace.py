"""Timeline of the lean upsampler (convtl_kernel, convtl_run's stamps; a library built with -DFV_PAIR_TRACE through
tools/build_variant.py + FV_AB_LIB) at the three shapes HiFi-GAN light runs it at batch 1, beside the launch's duration by
events and the same launch's duration in a hot loop: what a launch of ~1 us of matrix work spends its 10-20 us on
("or a committed trace showing why not").
    python tools/build_variant.py trace -DFV_PAIR_TRACE && FV_AB_LIB=fastvocoder_amd/libfv_trace.so python tools/convtl_trace.py"""
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

SHAPES = [(256, 128, 8, 1000), (128, 64, 5, 8000), (64, 32, 3, 40000)]      # HiFi-GAN light's upsamplers 1-3 (conf/hifigan/light.yaml)
names = ["A loads issued + window landed + merged + converted", "barrier", "K loop", "epilogue (stores issued)", "barrier"]
for cin, cout, s, T in SHAPES:
    k, pad, out_pad = 2 * s, s // 2 + s % 2, s % 2
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, cin, T), generator=g).to(dev)
    w = (torch.randn((cin, cout, k), generator=g) / (cin * 2) ** 0.5).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    P = _native.pack_conv_transpose1d_split(w, s)
    run = lambda: _native.conv_transpose1d_split_f16(x, P, bias, cout, k, s, pad, out_pad, pre_slope=0.1)  # noqa: E731
    for _ in range(300):                       # (sustained clocks: profiles/r06_clock_ramp.txt)
        y = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        run()
    e1.record()
    torch.cuda.synchronize()
    hot = e0.elapsed_time(e1) * 1e3 / 200
    trace.zero_()
    run()
    torch.cuda.synchronize()
    tr = trace.cpu().numpy()[:8 * nw * 8 * 16].reshape(8, nw, 8, 16)
    flop = 2.0 * T * cin * cout * 2 * s
    print(f"ConvTranspose1d {cin} -> {cout} x{s}, T = {T}, batch 1: {hot:.1f} us per launch in a hot loop (traced build); "
          f"{3 * flop / 2.5e15 * 1e6:.2f} us of matrix time at the f16 peak; {4.0 * (cin * T + cout * T * s) / 1e6:.1f} MB in + out")
    for blk in range(8):
        t12, t13, t14 = (int(v) for v in tr[blk, 0, 7, 12:15])
        if t12 == 0:
            continue
        line = f"   block {64 * blk:3d}: first window requested {t13 - t12}"
        for it in range(7):
            e = [int(v) for v in tr[blk, 0, it, :6]]
            if e[0] == 0:
                break
            if it == 0:
                line += f"; chunk 0 starts {e[0] - t12} cycles after the run"
            d = [e[1] - e[0] if e[1] else 0, (e[2] - e[1]) if e[1] else e[2] - e[0], e[3] - e[2], e[4] - e[3], (e[5] - e[4]) if e[5] else 0]
            line += f"\n      chunk {it}: " + "  ".join(f"{n} {v}" for n, v in zip(names, d))
        line += f"\n      run start -> stores drained: {t14 - t12} cycles"
        print(line)
    e = tr[0, :, 0, :5]
    print("   block 0, chunk 0, stamps 0..4 of every wave relative to wave 0's stamp 0:")
    for wv in range(nw):
        print(f"      wave {wv}: " + " ".join(f"{int(e[wv, i] - e[0, 0]):7d}" if e[wv, i] else "      -" for i in range(5)))
