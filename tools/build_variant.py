"""Build a VARIANT of the library for an A/B on one box (tuning aid): extra hipcc flags (-D switches of the kernels), its
own object directory, its own output file -- the product library is not touched.
    python tools/build_variant.py nolow "-DFV_NO_LOWGUARD"   ->  fastvocoder_amd/libfv_nolow.so
Use with tools/ab_lib.py or FV_AB_LIB=<path> python tools/pair_bench.py ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastvocoder_amd import _native  # noqa: E402

name, flags = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
os.environ["FV_HIPCC_FLAGS"] = flags
_native.LIB_PATH = os.path.join(ROOT, "fastvocoder_amd", f"libfv_{name}.so")
_native._HERE_BUILD = name
import subprocess  # noqa: E402
orig_join = os.path.join


def build():
    srcs = [os.path.join(_native._CSRC, s) for s in _native.SOURCES]
    want = _native.source_hash()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    fl = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f'-DFV_BUILD_ID="{want}"', "-Wno-unused-value", "-Wno-comment",
          "-Wno-pass-failed", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + flags.split()
    objdir = os.path.join(ROOT, "fastvocoder_amd", "build_" + name)
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        jobs.append((obj, subprocess.Popen([hipcc] + fl + ["-c", src, "-o", obj], stderr=subprocess.DEVNULL)))
    for obj, proc in jobs:
        assert proc.wait() == 0, obj
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _ in jobs] + ["-o", _native.LIB_PATH])
    print(_native.LIB_PATH)


build()
