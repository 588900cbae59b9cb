"""Repeat every BASELINE config's forward N times and compare each output with the first, bit for bit (a race in a kernel
shows up as a run that differs): python tools/soak.py [N = 200]"""
import os
import sys

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd.bin.synthesize import build_generator  # noqa: E402
from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
CONFIGS = [("melgan", "conf/melgan/original.yaml", 1, 200), ("hifigan", "conf/hifigan/light.yaml", 1, 1000),
           ("hifigan", "conf/hifigan/light.yaml", 3, 777), ("multiband-hifigan", "conf/multiband-hifigan/light.yaml", 4, 500),
           ("basis-melgan", "conf/basis-melgan/light.yaml", 1, 1000), ("basis-melgan", "conf/basis-melgan/light.yaml", 6, 333),
           ("hifigan", "conf/hifigan/large.yaml", 2, 400),
           # (batches large enough for the wide tiles: 256-column pairs at 64 channels, 128-column ones at 128, 64-column stacks)
           ("hifigan", "conf/hifigan/light.yaml", 16, 1000), ("basis-melgan", "conf/basis-melgan/light.yaml", 8, 1000),
           ("multiband-hifigan", "conf/multiband-hifigan/light.yaml", 16, 1000), ("hifigan", "conf/hifigan/large.yaml", 8, 600),
           # (round 5: the 32-channel one-launch stage with history -- 560 frames: one window per block, batch 4: six; the
           # transposed conv on resident images and the ring-free 256-channel convs at their batch sizes)
           ("hifigan", "conf/hifigan/light.yaml", 1, 560), ("hifigan", "conf/hifigan/light.yaml", 4, 1000),
           ("basis-melgan", "conf/basis-melgan/light.yaml", 32, 1000), ("hifigan", "conf/hifigan/large.yaml", 32, 1000)]
bad = 0
for name, path, B, T in CONFIGS:
    cfg = yaml.safe_load(open(path))
    m = build_generator(name, cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(name, cfg).items()})
    m = m.cuda().eval()
    m.remove_weight_norm()
    mel = torch.from_numpy(seeded_mel(T, seed=2, batch=B)).cuda()
    fn = (lambda: m.synthesize_batch(mel)) if name == "multiband-hifigan" else (lambda: m._samples(mel)) if name == "basis-melgan" else (lambda: m(mel))
    with torch.no_grad():
        first = fn().clone()
        diff = 0
        for _ in range(N if B * T <= 4000 else max(20, N // 10)):
            diff += int(not torch.equal(fn(), first))
    torch.cuda.synchronize()
    bad += diff
    print(f"{name:18s} {os.path.basename(path):14s} B={B} T={T}: {N if B * T <= 4000 else max(20, N // 10)} runs, {diff} differ from the first; "
          f"finite: {bool(torch.isfinite(first).all())}")
sys.exit(1 if bad else 0)
