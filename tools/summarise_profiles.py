"""Turn gpurun_out/<tag>/ (tools/collect_profiles.sh) into the committed profiles/<tag>_* summaries."""
import csv
import os
import shutil
import subprocess
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
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
    fams = {"split-f16 convs with streamed weights (convh_kernel, convp_kernel, convq_kernel, convq2_kernel)": ("convh_kernel", "convp_kernel", "convq_kernel", "convq2_kernel"), "split-f16 fused pairs (pairh_kernel)": ("pairh_kernel",),
            "split-f16 transposed convs (convt_kernel)": ("convt_kernel",),
            "fp32-MFMA convs (conv_mfma_kernel + conv_group3_kernel + conv_sum3_kernel + pair_kernel + pair_sum_kernel)":
                ("conv_mfma_kernel", "conv_group3_kernel", "conv_sum3_kernel", "fv::pair_kernel", "pair_sum_kernel")}
    fam = {name: defaultdict(float) for name in fams}
    for k in sorted(tot, key=lambda k: -tot[k].get("GRBM_GUI_ACTIVE", 0) * n[k].get("GRBM_GUI_ACTIVE", 0)):
        d, c = tot[k], n[k]
        f.write(k + "\n")
        for name in sorted(d):
            f.write(f"    {name:28s} {d[name] / c[name]:16.0f}   (n={c[name]})\n")
        if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_INSTS_MFMA"):
            f.write(f"    MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs) = "
                    f"{d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f};  "
                    f"VALU instructions per MFMA = {d['SQ_INSTS_VALU'] / max(d['SQ_INSTS_MFMA'], 1):.2f}\n")
        for name, pats in fams.items():
            if any(t in k for t in pats):
                for cn in d:
                    fam[name][cn] += d[cn]
    for name, d in fam.items():
        if not d.get("GRBM_GUI_ACTIVE"):
            continue
        f.write(f"{name}, all launches:\n")
        f.write(f"    MFMA-busy fraction of SIMD cycles = {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}\n")
        f.write(f"    VALU instructions per MFMA (incl. the MFMA itself) = {d['SQ_INSTS_VALU'] / max(d['SQ_INSTS_MFMA'], 1):.2f}\n")
        f.write(f"    SQ_LDS_BANK_CONFLICT total = {d['SQ_LDS_BANK_CONFLICT']:.0f}\n")
print(open(f"{dst}/{tag}_mfma_counters.txt").read()[-1200:])

# ---- the other BASELINE configs: kernel-time shares from rocprofv3 --stats ----
names = {0: "config 1: MelGAN original, T=200, B=1", 2: "config 3: MB-HiFi-GAN light + PQMF, T=1000, B=32",
         3: "config 4: Basis-MelGAN light, T=1000, B=64", 4: "config 5 (one GPU's share): HiFi-GAN large, T=1000, B=64"}
with open(f"{dst}/{tag}_configs.md", "w") as f:
    f.write(f"# {tag}: the other BASELINE configs on one MI355X (tools/collect_profiles.sh)\n\n")
    if os.path.exists(f"{src}/configs.log"):
        f.write("Throughput (tools/bench_configs.py, un-profiled):\n\n```\n" + open(f"{src}/configs.log").read() + "```\n\n")
    for i, title in names.items():
        path = f"{src}/cfg{i}/cfg_kernel_stats.csv"
        if not os.path.exists(path):
            continue
        shutil.copy(path, f"{dst}/{tag}_cfg{i}_kernel_stats.csv")
        rows = [r for r in csv.DictReader(open(path)) if "fv::" in r["Name"] and "pack" not in r["Name"]
                and "fold" not in r["Name"]]
        total = sum(float(r["TotalDurationNs"]) for r in rows)
        f.write(f"## {title}\n\nrocprofv3 --kernel-trace --stats, fv:: kernels of the forward passes "
                f"(profiles/{tag}_cfg{i}_kernel_stats.csv):\n\n| kernel | calls | avg us | share |\n|---|---|---|---|\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
            f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{100 * float(r['TotalDurationNs']) / total:.1f} % |\n")
        f.write("\n")
print(open(f"{dst}/{tag}_configs.md").read()[:1500])
