import sys, time, torch, yaml, numpy as np
sys.path.insert(0, '/root/repo')
from fastvocoder_amd.bin.synthesize import build_generator
from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict
for name, path in (("hifigan", "conf/hifigan/light.yaml"), ("melgan", "conf/melgan/original.yaml"), ("multiband-hifigan", "conf/multiband-hifigan/large.yaml"), ("basis-melgan", "conf/basis-melgan/light.yaml")):
    cfg = yaml.safe_load(open('/root/repo/' + path))
    m = build_generator(name, cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(name, cfg).items()})
    m = m.cuda().eval(); m.remove_weight_norm()
    for T in (30000, 70001):
        mel = seeded_mel(T, seed=3)
        y = m.inference(mel)                        # first call at this length: builds the plans of its chunk shapes
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = m.inference(mel)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        # spot-check a window in the middle against a short run with halo (exactness of chunking)
        a = T // 2 - 40
        ref = m.inference(mel[a - 64:a + 64 + 32])
        hop = 240
        off = 0
        if name == "multiband-hifigan":
            off = 40
        got = y[(a) * hop - off:(a + 32) * hop - off]
        want = ref[64 * hop - off:(64 + 32) * hop - off]
        err = float((got - want).abs().max())
        print(f"{name:18s} T={T:6d} -> {y.numel():9d} samples in {dt*1e3:8.1f} ms  finite={bool(torch.isfinite(y).all())}  mid-window vs short run: {err:.2e}  mem={torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
