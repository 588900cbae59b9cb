"""Throughput of HiFi-GAN light against the batch size (mel 80x1000 per utterance): python tools/batch_sweep.py"""
import os
import sys
import time

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd.bin.synthesize import build_generator  # noqa: E402
from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict  # noqa: E402

cfg = yaml.safe_load(open("conf/hifigan/light.yaml"))
m = build_generator("hifigan", cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict("hifigan", cfg).items()})
m = m.cuda().eval()
m.remove_weight_norm()
for B in (1, 2, 4, 8, 16, 32, 64):
    mel = torch.from_numpy(seeded_mel(1000, seed=1, batch=B)).cuda()
    with torch.no_grad():
        for _ in range(3):
            m(mel)
        torch.cuda.synchronize()
        n = max(3, 60 // B)
        t0 = time.perf_counter()
        for _ in range(n):
            m(mel)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f"B={B:3d}: {dt * 1e3:8.3f} ms/step  {B * 240000 / dt / 1e6:7.1f} Msamples/s  {dt * 1e3 / B:6.3f} ms per utterance")
