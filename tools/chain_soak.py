"""Soak of the chained stage launch (fv_tuning_set "chain" = 1): N forwards of HiFi-GAN light at batch 1 and batch 3, every
output compared bit for bit with the unchained one; the guard word must stay clear (no spin time-out, no stale tile).
    python tools/chain_soak.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fastvocoder_amd import _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
model, cfg, sd = bench.build_model("light", dev, None, 0)
mels = [torch.from_numpy(bench.utterance_mels(0, b)).to(dev) for b in (1, 3)]
with torch.no_grad():
    _native.tuning_set("chain", 0)
    refs = [model(m).clone() for m in mels]
    _native.tuning_set("chain", 1)
    bad = 0
    for i in range(n):
        for m, r in zip(mels, refs):
            if not torch.equal(model(m), r):
                bad += 1
    clean = not model.check_range()
_native.tuning_set("chain", 0)
print(f"{n} x 2 chained forwards: {bad} differing outputs, guard word clean: {clean}")
sys.exit(1 if bad or not clean else 0)
