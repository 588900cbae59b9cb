"""Host-side enqueue cost of one forward vs its GPU time (is the step launch-bound?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yaml
from fastvocoder_amd.bin.synthesize import build_generator
from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict
cfg = yaml.safe_load(open("conf/hifigan/light.yaml"))
m = build_generator("hifigan", cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict("hifigan", cfg).items()})
m = m.cuda().eval(); m.remove_weight_norm()
mel = torch.from_numpy(seeded_mel(1000, batch=1)).cuda()
with torch.no_grad():
    for _ in range(5): m(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): m(mel)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"host enqueue per forward: {(t1-t0)/20*1e3:.3f} ms; total per forward incl. GPU drain: {(t2-t0)/20*1e3:.3f} ms")
