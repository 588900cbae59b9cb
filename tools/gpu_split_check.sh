#!/bin/bash
# split-f16 pair kernels: parity tests, per-launch timing next to the fp32 kernels, one bench line
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pairs.py -x -q 2>&1 | tail -15 > gpurun_out/split_tests.log
timeout 300 python tools/pair_bench.py 16 240000 1 split 2>&1 | grep -v amdgpu.ids > gpurun_out/split_bench.log
timeout 300 python tools/pair_bench.py 16 240000 1 f32 2>&1 | grep -v amdgpu.ids | grep -v member >> gpurun_out/split_bench.log
timeout 300 python tools/pair_bench.py 32 120000 1 split 2>&1 | grep -v amdgpu.ids >> gpurun_out/split_bench.log
timeout 300 python tools/pair_bench.py 32 120000 1 f32 2>&1 | grep -v amdgpu.ids | grep -v member >> gpurun_out/split_bench.log
for d in 2 4 6 8; do FV_TUNING=1 FV_PAIR_DBG=$d timeout 300 python tools/pair_bench.py 32 120000 1 split 2>&1 | grep "pairs dil=5" >> gpurun_out/split_bench.log; done
for sk in 5 8; do FV_PAIRH_SKEL=$sk timeout 300 python tools/pair_bench.py 16 240000 1 split 2>&1 | grep "pairs dil" | sed "s/^/skel=$sk /" >> gpurun_out/split_bench.log; done
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/split_bench_line.json 2> gpurun_out/split_bench_line.err
FV_PAIR_PREC=f32 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/f32_bench_line.json 2>/dev/null
cat gpurun_out/split_tests.log gpurun_out/split_bench.log
python - <<'PY'
import json
for n in ("split","f32"):
    try:
        d=json.loads(open(f"gpurun_out/{n}_bench_line.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["parity"]["max_abs_vs_reference_golden"], d["roofline"]["by_family_ms_per_step"], d.get("roofline_hbm_stage",{}).get("frac"))
    except Exception as e: print(n, "failed", e)
PY
tail -5 gpurun_out/split_bench_line.err
