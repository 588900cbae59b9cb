"""Turn gpurun_out/<tag>/ (tools/collect_profiles.sh) into the committed profiles/<tag>_* summaries."""
import csv
import os
import shutil
import subprocess
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = f"gpurun_out/{tag}", "profiles"
os.makedirs(dst, exist_ok=True)
shutil.copy(f"{src}/stats/bench_kernel_stats.csv", f"{dst}/{tag}_bench_kernel_stats.csv")
shutil.copy(f"{src}/bench.json", f"{dst}/{tag}_bench.json")
subprocess.check_call([sys.executable, "tools/pmc_traffic.py", f"{src}/fetch/bench_counter_collection.csv",
                       f"{src}/write/bench_counter_collection.csv", f"{dst}/{tag}_hbm_traffic.json"],
                      stdout=subprocess.DEVNULL)
tot = defaultdict(lambda: defaultdict(float))
n = defaultdict(lambda: defaultdict(int))
for row in csv.DictReader(open(f"{src}/mfma/bench_counter_collection.csv")):
    k = row["Kernel_Name"]
    if "fv::" not in k:
        continue
    tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[k][row["Counter_Name"]] += 1
with open(f"{dst}/{tag}_mfma_counters.txt", "w") as f:
    f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY\n"
            "#   SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -- python bench.py --steps 5 --warmup 2\n"
            "# MI355X, HiFi-GAN light B=1 T=1000; per-dispatch averages; GRBM_GUI_ACTIVE is summed over the 8 XCDs\n")
    fam = defaultdict(float)
    for k in sorted(tot, key=lambda k: -tot[k].get("GRBM_GUI_ACTIVE", 0)):
        d, c = tot[k], n[k]
        f.write(k + "\n")
        for name in sorted(d):
            f.write(f"    {name:28s} {d[name] / c[name]:16.0f}   (n={c[name]})\n")
        if ("conv_mfma_kernel" in k or "conv_group3_kernel" in k or "conv_sum3_kernel" in k) and d.get("GRBM_GUI_ACTIVE"):
            f.write(f"    MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs) = "
                    f"{d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f};  "
                    f"VALU instructions per MFMA = {d['SQ_INSTS_VALU'] / max(d['SQ_INSTS_MFMA'], 1):.2f}\n")
            for name in d:
                fam[name] += d[name]
    f.write("conv family (conv_mfma_kernel + conv_group3_kernel + conv_sum3_kernel), all launches:\n")
    f.write(f"    MFMA-busy fraction of SIMD cycles = {fam['SQ_VALU_MFMA_BUSY_CYCLES'] / (fam['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}\n")
    f.write(f"    VALU instructions per MFMA (incl. the MFMA itself) = {fam['SQ_INSTS_VALU'] / fam['SQ_INSTS_MFMA']:.2f}\n")
    f.write(f"    SQ_LDS_BANK_CONFLICT total = {fam['SQ_LDS_BANK_CONFLICT']:.0f}\n")
print(open(f"{dst}/{tag}_mfma_counters.txt").read()[-700:])
