"""A/B of launcher tuning switches inside one process (tuning aid): HiFi-GAN light, B utterances of 1000 frames; every
argument is one setting "key=value,key=value" applied through fv_tuning_set on top of the defaults; prints the step time,
the 128- / 64-channel family times and whether the waveform equals the first setting's bit for bit.
    python tools/ab_tuning.py [--batch B] "" "pair128_unfused=1" "sched=2,sched_switch=0" ..."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fastvocoder_amd import _native  # noqa: E402

DEFAULTS = {"sched": 1, "sched_switch": 4, "convh_skel": -1, "convp_skel": 5, "convq_skel": -1, "pair128_unfused": 0,
            "convh_blocks": 0, "pair_blocks": 0, "pair_dbg": 0, "shape32": -1, "shape64": -1, "units": 500,
            "mrf_blocks": 0, "mrf_shape": 0, "convt_lean": 50, "convs_ringfree": -1, "convu_resident": 1}

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("settings", nargs="*", default=[""])
args = ap.parse_args()
dev = torch.device("cuda:0")
model, cfg, sd = bench.build_model("light", dev, None, 0)
mel = torch.from_numpy(bench.utterance_mels(0, args.batch)).to(dev)


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
    for name, kind in (("convh128", _native.KERNEL_CONVH128), ("convh64", _native.KERNEL_CONVH64),
                       ("pairh32", _native.KERNEL_PAIRH32), ("pairh16", _native.KERNEL_PAIRH16),
                       ("conv32", _native.KERNEL_CONV_MFMA32), ("convt", _native.KERNEL_CONVT)):
        r = _native.profile_collect(kind)
        out[name] = (round(1e3 * r["ms"] / reps, 1), r["launches"] // reps)
    _native.profile_collect(-1)
    return out


ref = None
for setting in args.settings:
    for k, v in DEFAULTS.items():
        _native.tuning_set(k, v)
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("=")
        _native.tuning_set(k, int(v))
    ms, y = run()
    ms2, _ = run()
    if ref is None:
        ref = y.clone()
    print(f"[{setting or 'defaults'}] {ms:.4f} / {ms2:.4f} ms/step  (us incl. event cost, launches) {families()}  "
          f"same_bits={bool(torch.equal(y, ref))}", flush=True)
