"""Whole HiFi-GAN-light forwards with the 32-channel MRF stage as ONE launch (fuse_stage = (16, 32)) against pair launches
(fuse_stage = (16,)), over utterance lengths and batch sizes: where does the one-launch kernel pay?  (The default policy,
fuse_stage = True, picks per call: hifigan._stage_one_launch.)  Interleaved rounds, min / median.
usage: python tools/stage_policy_bench.py [frames ...]"""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import cases  # noqa: E402
from fastvocoder_amd.bin.synthesize import build_generator  # noqa: E402
from fastvocoder_amd.synthetic import seeded_state_dict  # noqa: E402


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    frames = [int(v) for v in sys.argv[1:]] or [125, 250, 500, 560, 700, 1000]
    dev = torch.device("cuda:0")
    cfg = cases.load_conf("conf/hifigan/light.yaml")
    models = {}
    for name, fs in (("pairs", (16,)), ("one launch", (16, 32)), ("default", True)):
        m = build_generator("hifigan", cfg)
        sd = seeded_state_dict("hifigan", cfg, seed=0)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.to(dev).eval()
        m.remove_weight_norm()
        m.range_guard = "lazy"
        m.fuse_stage = fs
        models[name] = m
    for B in (1, 4):
        for T in frames:
            x = torch.randn(B, 80, T, device=dev)
            best = {k: [] for k in models}
            with torch.no_grad():
                for _ in range(5):
                    for name, m in models.items():
                        best[name].append(timed(lambda: m(x)))
            print(f"B={B} frames={T}: " + "  ".join(f"{k} {min(v):7.1f}/{sorted(v)[2]:7.1f}" for k, v in best.items())
                  + f"   tag {models['default']._flag_tag(T)}")
    for m in models.values():
        assert not m.check_range()


if __name__ == "__main__":
    main()
