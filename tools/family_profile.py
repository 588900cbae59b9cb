"""Per-family kernel time and algorithmic TFLOP/s of a forward at batch B (the measurement hook of bench.py):
python tools/family_profile.py [B] [generator = hifigan] [config = conf/hifigan/light.yaml] [T = 1000]"""
import os
import sys

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd import _native  # noqa: E402
from fastvocoder_amd.bin.synthesize import build_generator  # noqa: E402
from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
GEN = sys.argv[2] if len(sys.argv) > 2 else "hifigan"
PATH = sys.argv[3] if len(sys.argv) > 3 else "conf/hifigan/light.yaml"
T = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
cfg = yaml.safe_load(open(PATH))
m = build_generator(GEN, cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(GEN, cfg).items()})
m = m.cuda().eval()
m.remove_weight_norm()
mel = torch.from_numpy(seeded_mel(T, seed=1, batch=B)).cuda()
kinds = {"conv32": _native.KERNEL_CONV_MFMA32, "pairh16": _native.KERNEL_PAIRH16, "pairh32": _native.KERNEL_PAIRH32,
         "convh64": _native.KERNEL_CONVH64, "convh128": _native.KERNEL_CONVH128, "convt": _native.KERNEL_CONVT,
         "narrow": _native.KERNEL_CONV_NARROW, "conv16": _native.KERNEL_CONV_MFMA16, "convg": _native.KERNEL_CONVG,
         "stack": _native.KERNEL_STACK}
fwd = (lambda: m.synthesize_batch(mel)) if GEN == "multiband-hifigan" else (lambda: m(mel))
with torch.no_grad():
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    _native.profile_enable(True)
    reps = 3
    for _ in range(reps):
        fwd()
    torch.cuda.synchronize()
    _native.profile_enable(False)
tot = 0.0
for name, k in kinds.items():
    r = _native.profile_collect(k)
    if r["launches"]:
        tot += r["ms"] / reps
        print(f"B={B} {name:9s} {r['launches'] // reps:3d} launches {r['ms'] / reps:8.3f} ms  {r['flops'] / (r['ms'] * 1e-3) / 1e12:6.1f} TFLOP/s  "
              f"{r['bytes'] / (r['ms'] * 1e-3) / 1e9:7.0f} GB/s external")
print(f"B={B} total {tot:.3f} ms")
