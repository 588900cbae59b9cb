#!/bin/bash
# per-kernel average times of the bench step (rocprofv3 --kernel-trace --stats) beside bench.py's own HIP-event figures
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/ks; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks -o bench -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $R/gpurun_out/ks/bench.log 2>&1
cd $R; python - <<PY
import csv
tot=n=0
for r in csv.DictReader(open("gpurun_out/ks/bench_kernel_stats.csv")):
    if "fv::" in r["Name"] and "pack" not in r["Name"]:
        print(r["Name"][:72], r["Calls"], round(float(r["AverageNs"])/1e3,1))
        if "convh_kernel" in r["Name"] or "convp_kernel" in r["Name"]:
            tot+=float(r["TotalDurationNs"]); n+=int(r["Calls"])
print("rocprofv3: convh + convp average", round(tot/n/1e3,2), "us over", n, "dispatches")
PY
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('bench.py: ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'avg_launch_us', r['avg_launch_us'], r['measured'][-90:], 'kernel_ms', r['whole_step']['kernel_ms'], 'gaps', r['whole_step']['launch_gaps_ms']); print(r['by_family_ms_per_step']); print(d['roofline_hbm_stage']['frac'])"
