"""Where the HOST time of one forward goes (cProfile over 200 forwards, GPU drained only at the end):
python tools/host_profile.py [generator = melgan] [config = conf/melgan/original.yaml] [T = 200]"""
import cProfile
import os
import pstats
import sys
import time

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd.bin.synthesize import build_generator  # noqa: E402
from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict  # noqa: E402

GEN = sys.argv[1] if len(sys.argv) > 1 else "melgan"
PATH = sys.argv[2] if len(sys.argv) > 2 else "conf/melgan/original.yaml"
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cfg = yaml.safe_load(open(PATH))
m = build_generator(GEN, cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(GEN, cfg).items()})
m = m.cuda().eval()
m.remove_weight_norm()
mel = torch.from_numpy(seeded_mel(T, seed=1, batch=1)).cuda()
with torch.no_grad():
    for _ in range(10):
        m(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        m(mel)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3 * (t1 - t0) / 200:.3f} ms per forward; with the GPU drained {1e3 * (t2 - t0) / 200:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        m(mel)
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
