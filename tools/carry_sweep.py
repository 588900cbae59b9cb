"""A/B of the block schedule and the member hand-over of the weight-streaming kernels (tuning aid):
HiFi-GAN light, B utterances of 1000 frames; for every (convh_carry, sched, sched_switch) setting the step time, the
per-family kernel time and whether the waveform is bit-identical to the default's.  python tools/carry_sweep.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fastvocoder_amd import _native  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
model, cfg, sd = bench.build_model("light", dev, None, 0)
mel = torch.from_numpy(bench.utterance_mels(0, B)).to(dev)


def run(steps=60):
    with torch.no_grad():
        for _ in range(5):
            y = model(mel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = model(mel)
        torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, y


def families(reps=5):
    _native.profile_enable(True)
    with torch.no_grad():
        for _ in range(reps):
            model(mel)
    torch.cuda.synchronize()
    _native.profile_enable(False)
    out = {}
    for name, kind in (("convh128", _native.KERNEL_CONVH128), ("convh64", _native.KERNEL_CONVH64)):
        r = _native.profile_collect(kind)
        out[name] = round(1e3 * r["ms"] / reps, 1)
    _native.profile_collect(-1)
    return out


ref = None
for carry, sched, sw in ((1, 1, 4), (0, 1, 4), (1, 2, 4), (1, 2, 2), (1, 2, 1), (1, 2, 0), (0, 2, 4), (1, 0, 4), (0, 0, 4)):
    _native.tuning_set("convh_carry", carry)
    _native.tuning_set("sched", sched)
    _native.tuning_set("sched_switch", sw)
    ms, y = run()
    ms2, _ = run()
    if ref is None:
        ref = y.clone()
    print(f"carry={carry} sched={sched} switch={sw}: {ms:.4f} / {ms2:.4f} ms/step  families(us incl. event cost) {families()}"
          f"  same_bits={bool(torch.equal(y, ref))}", flush=True)
