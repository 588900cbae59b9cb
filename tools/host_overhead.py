"""Host time per forward (enqueue only, no synchronisation inside the loop) against the GPU-bound step time:
which configs are bound by the launch path?  python tools/host_overhead.py"""
import os
import sys
import time

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd.bin.synthesize import build_generator  # noqa: E402
from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict  # noqa: E402

for name, path, T in (("melgan", "conf/melgan/original.yaml", 200), ("hifigan", "conf/hifigan/light.yaml", 1000),
                      ("hifigan", "conf/hifigan/light.yaml", 100)):
    cfg = yaml.safe_load(open(path))
    m = build_generator(name, cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(name, cfg).items()})
    m = m.cuda().eval()
    m.remove_weight_norm()
    mel = torch.from_numpy(seeded_mel(T, seed=1, batch=1)).cuda()
    with torch.no_grad():
        for _ in range(5):
            m(mel)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            m(mel)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        plan = next(iter(m._fv_plans.values()))[1]
        from fastvocoder_amd import _native
        nl = _native.lib().fv_plan_num_ops(plan._h)
    print(f"{name} T={T}: host {1e3 * (t1 - t0) / n:.3f} ms per forward ({nl} plan ops), step {1e3 * (t2 - t0) / n:.3f} ms")
