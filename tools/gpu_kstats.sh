#!/bin/bash
# per-kernel average times of the bench step (rocprofv3 --kernel-trace --stats), printed
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/ks; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks -o bench -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $R/gpurun_out/ks/bench.log 2>&1
cd $R; python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ks/bench_kernel_stats.csv")):
    if "fv::" in r["Name"] and "pack" not in r["Name"]: print(r["Name"][:72], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
tail -1 gpurun_out/ks/bench.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['roofline']['frac'], d['roofline']['split_f16_transposed_convs'])"
