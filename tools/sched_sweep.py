"""A/B of the block schedule of the weight-streaming kernels (tuning aid): HiFi-GAN light, B utterances of 1000
frames; for every (sched, sched_switch, convh_skel, convp_skel) setting the step time, the per-family kernel time and
whether the waveform is bit-identical to the default's.  python tools/sched_sweep.py [B]"""
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
settings = [(1, 4, -1, 5)]
for sched in (2,):
    for sw in (0, 4):
        for hs, ps in ((-1, 5), (6, 8), (10, 12), (14, 16)):
            settings.append((sched, sw, hs, ps))
settings += [(1, 4, 6, 8), (1, 4, 10, 12), (0, 4, 6, 8), (0, 4, 10, 12)]
for sched, sw, hs, ps in settings:
    _native.tuning_set("sched", sched)
    _native.tuning_set("sched_switch", sw)
    _native.tuning_set("convh_skel", hs)
    _native.tuning_set("convp_skel", ps)
    ms, y = run()
    ms2, _ = run()
    if ref is None:
        ref = y.clone()
    print(f"sched={sched} switch={sw} convh_skel={hs} convp_skel={ps}: {ms:.4f} / {ms2:.4f} ms/step  "
          f"families(us incl. event cost) {families()}  same_bits={bool(torch.equal(y, ref))}", flush=True)
