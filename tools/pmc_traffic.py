"""HBM traffic of the conv kernel family from two rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE collected separately, MI355X_MICROARCH.md "HBM"):
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   per dispatch
FETCH_SIZE is doubled because on gfx950 this rocprofv3 reports exactly half the
bytes of a wide coalesced read (128-byte requests tallied at 64 B); WRITE_SIZE
is taken as reported (uncalibrated per the guide).  Counter unit: KiB.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            tot[row["Kernel_Name"]] += float(row["Counter_Value"])
            n[row["Kernel_Name"]] += 1
    return tot, n


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"unit": "bytes per launch", "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024", "kernels": {}}
    fam_bytes, fam_n = 0.0, 0
    split = {"16": [0.0, 0], "32": [0.0, 0]}          # fv::pairh_kernel<MH = 1 | 2, ...>: C = 16 | 32
    wide = [0.0, 0]                                   # fv::convh_kernel / convq2_kernel: the 128- / 64-channel stages
    stage16 = [0.0, 0]                                # fv::mrfh_kernel: the 16-channel stage as one launch
    for k in sorted(fetch, key=lambda k: -fetch[k]):
        if k not in write or "fv::" not in k:
            continue
        fb = 2.0 * fetch[k] / nf[k] * 1024
        wb = write[k] / nw[k] * 1024
        out["kernels"][k] = {"launches": nf[k], "fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes": fb + wb}
        if any(t in k for t in ("conv_mfma_kernel", "conv_group3_kernel", "conv_sum3_kernel", "fv::pair_kernel",
                                "pair_sum_kernel")):
            fam_bytes += (fb + wb) * nf[k]
            fam_n += nf[k]
        if "convh_kernel<" in k or "convp_kernel<" in k or "convq_kernel<" in k or "convq2_kernel<" in k:
            wide[0] += (fb + wb) * nf[k]
            wide[1] += nf[k]
        if "mrfh_kernel<" in k:
            stage16[0] += (fb + wb) * nf[k]
            stage16[1] += nf[k]
        if "pairh_kernel<" in k:
            c = "16" if "pairh_kernel<1," in k else "32"
            split[c][0] += (fb + wb) * nf[k]
            split[c][1] += nf[k]
    # the fp32-MFMA family bench.py's `roofline` is about; the split-f16 fused pairs per channel count
    out["conv_mfma_family"] = {"launches": fam_n, "hbm_bytes_per_launch": fam_bytes / max(fam_n, 1)}
    if wide[1]:
        out["split_f16_convs"] = {"launches": wide[1], "hbm_bytes_per_launch": wide[0] / wide[1]}
    if stage16[1]:
        out["split_f16_stage_c16"] = {"launches": stage16[1], "hbm_bytes_per_launch": stage16[0] / stage16[1]}
    for c, (b, n) in split.items():
        if n:
            out["split_f16_pairs_c" + c] = {"launches": n, "hbm_bytes_per_launch": b / n}
    js = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(js)
    print(js[:3000])


if __name__ == "__main__":
    main()
