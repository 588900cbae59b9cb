"""Registers, spills, scratch and LDS of every kernel in libfastvocoder_hip.so (from the code object's metadata notes).

    python tools/kernel_resources.py [filter]

Runs without a GPU: it reads the .hip_fatbin sections of the .o files under fastvocoder_amd/build."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as d:
        co, fb = os.path.join(d, "dev.co"), os.path.join(d, "fb.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj], capture_output=True)
        if not os.path.exists(fb):
            return []
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            return []
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = []
    for blk in notes.split("- .agpr_count")[1:]:
        def f(key):
            m = re.search(rf"\.{key}:\s+(\S+)", blk)
            return m.group(1) if m else "?"
        name = subprocess.run(["c++filt", f("name")], capture_output=True, text=True).stdout.strip()
        out.append((name, f("vgpr_count"), f("vgpr_spill_count"), f("sgpr_spill_count"), f("private_segment_fixed_size"),
                    f("group_segment_fixed_size")))
    return out


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    print(f"{'vgpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7}  kernel")
    for obj in sorted(glob.glob(os.path.join(ROOT, "fastvocoder_amd", "build", "*.o"))):
        for name, vg, vs, ss, scr, lds in kernels_of(obj):
            short = re.sub(r"^void ", "", name).split("(")[0]
            if flt in short:
                print(f"{vg:>5} {vs:>6} {ss:>6} {scr:>7}  {short}")


if __name__ == "__main__":
    main()
