"""Host time of one HiFi-GAN light forward, split three ways (no GPU drain inside the loops):
  forward()            module call: policy / plan lookup + Plan.run
  Plan.run             shape query, output allocation, ctypes marshalling, fv_plan_run_aux
  fv_plan_run_aux      the C call alone with prebuilt arguments
python tools/host_split.py [T = 1000] [reps = 300]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fastvocoder_amd import _native  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
bench.T_FRAMES = T
model, cfg, sd = bench.build_model("light", dev, None, 0)
mel = torch.from_numpy(bench.utterance_mels(0, 1)).to(dev)


def loop(fn, reps=REPS):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / reps, 1e6 * (t2 - t0) / reps


with torch.no_grad():
    for _ in range(5):
        model(mel)
    plan = model._trunk_plan(T)
    out = plan.run(mel)
    a = loop(lambda: model(mel))
    b = loop(lambda: plan.run(mel))
    b2 = loop(lambda: plan.run(mel, out=out))
    L = _native.lib()
    stream = torch.cuda.current_stream().cuda_stream
    ptrs = (ctypes.c_void_p * 2)(None, None)
    batched = (ctypes.c_int * 2)(0, 0)
    args = (plan._h, 1, T, mel.data_ptr(), out.data_ptr(), None, ptrs, batched, plan._ws.data_ptr(), plan._ws.numel(), stream)
    c = loop(lambda: L.fv_plan_run_aux(*args))
    e = loop(lambda: torch.empty((1, 1, 240 * T), dtype=torch.float32, device=dev))
    model.range_guard = "auto"
    d = loop(lambda: model(mel), 100)
print(f"T = {T}: host us per forward (enqueue only / with the GPU drained at the end)")
print(f"  forward() lazy      {a[0]:8.1f} / {a[1]:8.1f}")
print(f"  Plan.run            {b[0]:8.1f} / {b[1]:8.1f}")
print(f"  Plan.run(out=)      {b2[0]:8.1f} / {b2[1]:8.1f}")
print(f"  fv_plan_run_aux     {c[0]:8.1f} / {c[1]:8.1f}")
print(f"  torch.empty         {e[0]:8.1f}")
print(f"  forward() auto      {d[0]:8.1f} / {d[1]:8.1f}")
