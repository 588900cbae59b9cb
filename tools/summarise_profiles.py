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
    fams = {"split-f16 convs / fused pairs at 64+ channels (convh_kernel, convs_kernel, convq2_kernel)": ("convh_kernel", "convs_kernel", "convq2_kernel"),
            "split-f16 fused pairs at 32 channels (pairh_kernel)": ("pairh_kernel",),
            "split-f16 one-launch 16-channel MRF stage (mrfh_kernel)": ("mrfh_kernel",),
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

# ---- matrix-cycle accounting at saturation: per kernel family, executed MFMA FLOP (SQ_INSTS_MFMA x the
# FLOP of one v_mfma_f32_16x16x32_f16: 16 384) next to 3 x the algorithmic FLOP of the same forward, and MFMA-busy ----
import json  # noqa: E402

FAMILY_OF = [("mrfh_kernel", "mrf16"), ("mrfw_kernel", "mrf32"), ("pairh_kernel<1,", "pairh16"), ("pairh_kernel<2,", "pairh32"),
             ("convq2_kernel<1, 64>", "convh64"), ("convq2_kernel<3, 64>", "convh64"), ("convq2_kernel<5, 64>", "convh64"),
             ("convq2_kernel<1, 65>", "convh64"), ("convq2_kernel<3, 65>", "convh64"), ("convq2_kernel<5, 65>", "convh64"),
             ("convq2_kernel", "convh128"), ("convh_kernel<2,", "convh64"), ("convh_kernel", "convh128"), ("convs_kernel", "convh128"), ("convs2_kernel", "convh128"), ("convtl_kernel", "convt"), ("convu2_kernel", "convt"),
             ("convt_kernel", "convt"), ("convtn_kernel", "convt"), ("convu_kernel", "convt"), ("convg_kernel", "convg"),
             ("convr_kernel", "convg"), ("convk2_kernel", "stack"), ("convk_kernel", "stack")]
titles = {5: "HiFi-GAN light, 16 utterances of 1000 frames", 2: "config 3: MB-HiFi-GAN light + PQMF, batch 32",
          3: "config 4: Basis-MelGAN light, batch 64", 4: "config 5 (one GPU's share): HiFi-GAN large, batch 64"}
out_path = f"{dst}/{tag}_saturated_counters.txt"
with open(out_path, "w") as f:
    f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA\n"
            "#   --kernel-trace -- python tools/bench_configs.py --only i --steps 1 --families-json ...   (tools/collect_profiles.sh)\n"
            "# executed = SQ_INSTS_MFMA x 16 384 FLOP (one v_mfma_f32_16x16x32_f16 per wave), all dispatches of the family in the run;\n"
            "# algorithmic = the library's measurement hook for ONE forward (2 B Cout Cin k T per conv) x forwards in the run;\n"
            "# a split-f16 kernel executes THREE f16 FLOP per algorithmic FLOP by construction: executed / (3 x algorithmic) = 1 is ideal,\n"
            "# the excess is recomputed halo columns, the zero tap that pads an odd tap count, padded rows / chunks.\n")
    for i, title in titles.items():
        cpath, jpath = f"{src}/sat{i}/sat_counter_collection.csv", f"{src}/sat{i}_families.json"
        if not (os.path.exists(cpath) and os.path.exists(jpath)):
            continue
        fams = list(json.load(open(jpath)).values())[0]["families"]
        ktot = defaultdict(lambda: defaultdict(float))
        kn = defaultdict(int)
        for row in csv.DictReader(open(cpath)):
            k = row["Kernel_Name"]
            if "fv::" not in k or "pack" in k or "fold" in k or "row_scale" in k:
                continue
            ktot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "SQ_INSTS_MFMA":
                kn[k] += 1
        ftot, fdisp = defaultdict(lambda: defaultdict(float)), defaultdict(int)
        kdur, kdn = defaultdict(float), defaultdict(int)
        tpath = f"{src}/sat{i}/sat_kernel_trace.csv"
        if os.path.exists(tpath):
            for row in csv.DictReader(open(tpath)):
                kdur[row["Kernel_Name"]] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                kdn[row["Kernel_Name"]] += 1
        f.write(f"\n## {title}\n")
        for k in sorted(ktot, key=lambda k: -ktot[k].get("GRBM_GUI_ACTIVE", 0)):
            d = ktot[k]
            fam = next((fm for pat, fm in FAMILY_OF if pat in k), None)
            if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_INSTS_MFMA"):
                f.write(f"{k[:84]:84s} n={kn[k]:3d}  MFMA-busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}  "
                        f"VALU per MFMA {d['SQ_INSTS_VALU'] / d['SQ_INSTS_MFMA']:.2f}  executed {d['SQ_INSTS_MFMA'] * 16384 / 1e9 / max(kn[k], 1):9.1f} GFLOP / dispatch"
                        f"  busy cycles per MFMA {d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_INSTS_MFMA']:.1f}"
                        + (f"  clock {(d['GRBM_GUI_ACTIVE'] / 8 / kn[k]) / (kdur[k] / kdn[k]):.2f} GHz" if kdn.get(k) else "")
                        + f"  [{fam}]\n")
                if fam and kdn.get(k):
                    ftot[fam]["_ns"] += kdur[k]
            if fam:
                for cn, v in d.items():
                    ftot[fam][cn] += v
                fdisp[fam] += kn[k]
        f.write("family      dispatches  forwards  executed f16 GFLOP   3 x algorithmic   executed / (3 x alg)   MFMA-busy   VALU per MFMA   clock GHz   executed / peak at 2.4 GHz\n")
        for fam, d in ftot.items():
            if fam not in fams or not d.get("SQ_INSTS_MFMA"):
                continue
            forwards = fdisp[fam] / max(fams[fam]["launches"], 1)
            ex, alg3 = d["SQ_INSTS_MFMA"] * 16384 / 1e9, 3 * fams[fam]["flops"] * forwards / 1e9
            f.write(f"{fam:10s} {fdisp[fam]:10d} {forwards:9.1f} {ex:18.1f} {alg3:17.1f} {ex / alg3:22.3f} "
                    f"{d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024):11.3f} {d['SQ_INSTS_VALU'] / d['SQ_INSTS_MFMA']:15.2f}"
                    + (f" {d['GRBM_GUI_ACTIVE'] / 8 / d['_ns']:11.2f} {ex * 1e9 / (d['_ns'] * 1e-9) / 2.5e15:18.3f}" if d.get("_ns") else "") + "\n")
    f.write("\n# Reading (VERDICT r4 item 3: MFMA-busy 0.60 against 0.42 of the peak in algorithmic FLOP at 128 channels).  The counter\n"
            "# charges exactly 16 busy cycles per v_mfma_f32_16x16x32_f16, and executed / (3 x algorithmic) is 1.03-1.06 for the 64- / 128-channel\n"
            "# pairs: the busy matrix cycles ARE the algorithmic FLOP (plus 3-6 % halo columns / zero taps).  What separates 0.60 from 0.42 is\n"
            "# the CLOCK: under these MFMA-dense launches the chip runs at 1.8-1.9 GHz (GRBM_GUI_ACTIVE / dispatch duration, column above),\n"
            "# while the 2.5 PFLOP/s peak the fraction is priced against is the 2.4 GHz figure: 0.60 busy x 1.87 / 2.4 = 0.47 of that peak\n"
            "# executed = 0.44 algorithmic x 1.06.  (MI355X_MICROARCH.md, DVFS give-back: denser bodies clock lower; profiled passes\n"
            "# run another 2-3 % below un-profiled ones.)\n")
print(open(out_path).read()[:3000])
