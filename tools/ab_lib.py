"""A/B of two BUILDS of the library on one box (tuning aid): HiFi-GAN light, B utterances of 1000 frames, each library in
its own subprocess, the runs interleaved (A B A B ...) so that clock / thermal drift hits both alike.

    python tools/ab_lib.py [--batch B] [--rounds R] fastvocoder_amd/libfv_base_r3.so fastvocoder_amd/libfastvocoder_hip.so

A library name followed by "@nomerge" runs with NativeModule.merge_in_upsampler = False.
Prints per library the step time of every round, their median, and the per-family kernel times (profile hooks).
The product never loads a library by path from the environment: this tool sets _native.LIB_PATH in its own child
process, and does not go through bench.py's build-id check (an older build is the point)."""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib, batch, model_name, nomerge=False):
    sys.path.insert(0, ROOT)
    import torch
    from fastvocoder_amd import _native
    _native.LIB_PATH = os.path.abspath(lib)
    import ctypes
    has_merge = hasattr(ctypes.CDLL(_native.LIB_PATH), "fv_plan_set_input_merge")

    class Tolerant(ctypes.CDLL):            # an older build lacks the newest entry points: bind a stub that is never called
        def __getattr__(self, name):
            try:
                return super().__getattr__(name)
            except AttributeError:
                if not name.startswith("fv_"):
                    raise
                stub = ctypes.CFUNCTYPE(ctypes.c_int)(lambda: -1)
                setattr(self, name, stub)
                return stub
    _native.ctypes.CDLL = Tolerant
    import bench
    from fastvocoder_amd.generator.engine import NativeModule
    if nomerge or not has_merge:
        NativeModule.merge_in_upsampler = False          # (an older build has no merged upsampler input)
    dev = torch.device("cuda:0")
    model, cfg, sd = bench.build_model(model_name, dev, None, 0)
    mel = torch.from_numpy(bench.utterance_mels(0, batch)).to(dev)
    with torch.no_grad():
        for _ in range(8):
            y = model(mel)
        torch.cuda.synchronize()
        steps = 100 if batch == 1 else 10
        t0 = time.perf_counter()
        for _ in range(steps):
            y = model(mel)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        _native.profile_enable(True)
        for _ in range(5):
            model(mel)
        torch.cuda.synchronize()
        _native.profile_enable(False)
    fam = {}
    for name, kind in (("convh128", _native.KERNEL_CONVH128), ("convh64", _native.KERNEL_CONVH64),
                       ("pairh32", _native.KERNEL_PAIRH32), ("pairh16", _native.KERNEL_PAIRH16),
                       ("conv32", _native.KERNEL_CONV_MFMA32), ("convt", _native.KERNEL_CONVT)):
        r = _native.profile_collect(kind)
        fam[name] = round(1e3 * r["ms"] / 5, 1)
    print("ABLIB " + json.dumps({"ms": ms, "fam": fam, "sum": float(y.double().sum())}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--model", default="light")
    ap.add_argument("--child", default=None)
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    if a.child:
        return child(a.child.split("@")[0], a.batch, a.model, nomerge=a.child.endswith("@nomerge"))
    res = {lib: [] for lib in a.libs}
    for _ in range(a.rounds):
        for lib in a.libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, "--batch", str(a.batch),
                                  "--model", a.model], capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("ABLIB ")]
            if not line:
                print(lib, "FAILED", out.stderr[-600:])
                continue
            res[lib].append(json.loads(line[-1][6:]))
    for lib, rs in res.items():
        if not rs:
            continue
        ms = [r["ms"] for r in rs]
        fam = {k: statistics.median(r["fam"][k] for r in rs) for k in rs[0]["fam"]}
        print(f"{os.path.basename(lib):32s} median {statistics.median(ms):.4f} ms  runs {[round(m, 4) for m in ms]}  us/family {fam}  sum {rs[-1]['sum']:.6f}")


if __name__ == "__main__":
    main()
