#!/bin/bash
# split-f16 wide-channel pairs (convh): parity tests, per-launch timing, one bench line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pairs.py -x -q -k "wide" 2>&1 | tail -25 > gpurun_out/wide_tests.log
cat gpurun_out/wide_tests.log
timeout 300 python tools/pair_bench.py 64 40000 1 split 2>&1 | grep -v amdgpu.ids > gpurun_out/wide_bench.log
timeout 300 python tools/pair_bench.py 128 8000 1 split 2>&1 | grep -v amdgpu.ids >> gpurun_out/wide_bench.log
cat gpurun_out/wide_bench.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/wide_bench_line.json 2> gpurun_out/wide_bench_line.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/wide_bench_line.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["parity"]["max_abs_vs_reference_golden"], d["roofline"]["by_family_ms_per_step"])
except Exception as e: print("bench failed", e)
PY
tail -5 gpurun_out/wide_bench_line.err
