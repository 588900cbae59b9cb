"""One process, one library build (FV_AB_LIB, else the tree's): HiFi-GAN light forwards at batch 1 -- step time (min / median
of several timed loops) and the per-family kernel times of bench.py's accounting.  Run it alternately with two builds
(tools/build_variant.py) to compare them on one box.   [FV_AB_LIB=...] python tools/forward_ab.py [frames [batch]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fastvocoder_amd import _native  # noqa: E402
import _ablib  # noqa: E402
_ablib.use_lib_from_env(_native)
from fastvocoder_amd.bin.synthesize import build_generator  # noqa: E402
from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict  # noqa: E402
sys.path.insert(0, "tests")
import cases  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
cfg = cases.load_conf("conf/hifigan/light.yaml")
m = build_generator("hifigan", cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict("hifigan", cfg, seed=0).items()})
m = m.to(dev).eval()
m.remove_weight_norm()
m.range_guard = "lazy"
x = torch.from_numpy(seeded_mel(T, seed=1, batch=B)).to(dev)
ts = []
with torch.no_grad():
    for _ in range(5):
        m(x)
    for _ in range(7):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            m(x)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 30 * 1e3)
    kinds = {"convt": _native.KERNEL_CONVT, "convh64": _native.KERNEL_CONVH64, "convh128": _native.KERNEL_CONVH128,
             "pairh32": _native.KERNEL_PAIRH32, "mrf16": _native.KERNEL_MRF16, "conv32": _native.KERNEL_CONV_MFMA32}
    _native.profile_collect(-1)
    _native.profile_enable(True)
    for _ in range(20):
        m(x)
    torch.cuda.synchronize()
    _native.profile_enable(False)
    fam = {k: _native.profile_collect(v)["ms"] / 20 * 1e3 for k, v in kinds.items()}
assert not m.check_range()
print(f"{os.environ.get('FV_AB_LIB', 'tree'):40s} forward {min(ts):7.1f} / {sorted(ts)[3]:7.1f} us   " +
      " ".join(f"{k} {v:6.1f}" for k, v in fam.items()))
