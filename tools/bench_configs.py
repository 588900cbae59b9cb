"""Throughput of the other BASELINE.json configs (parity-test cases, not the headline bench):
    python tools/bench_configs.py [--steps K]
prints one line per config: samples/s, RTF@22.05k, ms/step, algorithmic TFLOP/s."""
import argparse
import os
import sys
import time

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvocoder_amd import _native  # noqa: E402
import _ablib  # noqa: E402
_ablib.use_lib_from_env(_native)       # FV_AB_LIB=<a library build>: A/B of two builds (tools/build_variant.py)
from fastvocoder_amd.bin.synthesize import build_generator  # noqa: E402
from fastvocoder_amd.synthetic import seeded_mel, seeded_state_dict  # noqa: E402

CONFIGS = [
    ("melgan T=200 B=1 (config 1)", "melgan", "conf/melgan/original.yaml", 1, 200),
    ("hifigan light T=1000 B=1 (config 2)", "hifigan", "conf/hifigan/light.yaml", 1, 1000),
    ("mb-hifigan light +PQMF T=1000 B=32 (config 3)", "multiband-hifigan", "conf/multiband-hifigan/light.yaml", 32, 1000),
    ("basis-melgan light T=1000 B=64 (config 4)", "basis-melgan", "conf/basis-melgan/light.yaml", 64, 1000),
    ("hifigan large T=1000 B=64 (config 5, one GPU's share)", "hifigan", "conf/hifigan/large.yaml", 64, 1000),
    ("hifigan light T=1000 B=16", "hifigan", "conf/hifigan/light.yaml", 16, 1000),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--only", type=int, default=None, help="index into CONFIGS (0-based): run just that one")
    ap.add_argument("--tuning", default="", help='launcher switches "key=value,..." (fv_tuning_set) for an A/B')
    ap.add_argument("--no-merge", action="store_true", help="A/B: the MRF merge in the stage's own last launch (merge_in_upsampler = False)")
    ap.add_argument("--batch", type=int, default=None, help="override the batch size of the selected configs")
    ap.add_argument("--no-last-stack", action="store_true", help="A/B: the graph's last ResidualStack as two launches (fuse_last = False)")
    ap.add_argument("--no-stack", action="store_true", help="A/B: MelGAN's ResidualStacks as two launches each (fuse_stack = False)")
    ap.add_argument("--families-json", default=None, help="write {family: {launches, algorithmic flops}} of ONE forward of each "
                    "selected config to this file (the matrix-cycle accounting of tools/summarise_profiles.py)")
    args = ap.parse_args()
    fam_out = {}
    if args.no_last_stack:
        from fastvocoder_amd.generator.modules import ResidualStack
        ResidualStack.fuse_last = False
    if args.no_stack:
        from fastvocoder_amd.generator.modules import ResidualStack
        ResidualStack.fuse_stack = False
    if args.no_merge:
        from fastvocoder_amd.generator.engine import NativeModule
        NativeModule.merge_in_upsampler = False
    for kv in filter(None, args.tuning.split(",")):
        k, v = kv.split("=")
        _native.tuning_set(k, int(v))
    dev = torch.device("cuda:0")
    for idx, (label, name, path, B, T) in enumerate(CONFIGS):
        if args.only is not None and idx != args.only:
            continue
        if args.batch is not None:
            B, label = args.batch, f"{label} [B={args.batch}]"
        cfg = yaml.safe_load(open(path))
        m = build_generator(name, cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(name, cfg).items()})
        m = m.to(dev).eval()
        m.remove_weight_norm()
        m.range_guard = "lazy"           # stream-ordered steps (the module default checks every call before it returns)
        if "pair_dbg" in args.tuning:
            m.range_guard = "off"        # (ablation switches give wrong values: timing only)
        mel = torch.from_numpy(seeded_mel(T, seed=1, batch=B)).to(dev)
        if name == "multiband-hifigan":
            fn = lambda: m.synthesize_batch(mel)
        elif name == "basis-melgan":
            fn = lambda: m._samples(mel)
        else:
            fn = lambda: m(mel)
        with torch.no_grad():
            y = fn()
            torch.cuda.synchronize()
            _native.profile_enable(True)
            fn()
            torch.cuda.synchronize()
            _native.profile_enable(False)
            if args.families_json:
                kinds = {"conv32": _native.KERNEL_CONV_MFMA32, "conv16": _native.KERNEL_CONV_MFMA16, "pairh16": _native.KERNEL_PAIRH16,
                         "pairh32": _native.KERNEL_PAIRH32, "convh64": _native.KERNEL_CONVH64, "convh128": _native.KERNEL_CONVH128,
                         "convt": _native.KERNEL_CONVT, "convg": _native.KERNEL_CONVG, "stack": _native.KERNEL_STACK,
                         "mrf16": _native.KERNEL_MRF16, "mrf32": _native.KERNEL_MRF32}
                rec = {k: _native.profile_collect(v) for k, v in kinds.items()}
                fam_out[label] = {"batch": B, "frames": T,
                                  "families": {k: {"launches": int(r["launches"]), "flops": r["flops"]} for k, r in rec.items() if r["launches"]}}
            prof = _native.profile_collect(-1) if not args.families_json else {"flops": sum(r["flops"] for r in rec.values())}
            t0 = time.perf_counter()
            steps = args.steps * (20 if B * T <= 2000 else 1)      # (a 0.3 ms step: 5 of them time the host's wake-up)
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        n = y.numel()
        print(f"{label:52s} {n/dt/1e6:9.1f} Msamples/s  RTF@22.05k {dt/(n/22050):.2e}  {dt*1e3:9.3f} ms/step  "
              f"{prof['flops']/dt/1e12:6.1f} TFLOP/s algorithmic", flush=True)
        del m, mel, y
        torch.cuda.empty_cache()
    if args.families_json:
        import json
        with open(args.families_json, "w") as f:
            json.dump(fam_out, f, indent=1)


if __name__ == "__main__":
    main()
