set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2_gpu.log
timeout 300 python bench.py --steps 30 --warmup 5 > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
cat gpurun_out/r2_gpu.log; tail -3 gpurun_out/r2_bench1.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2_bench1.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("ms_per_step", d["ms_per_step"], "frac", r["frac"], "achieved", r["achieved"], "peak", r["peak"], "whole", r["whole_step"], "kernel_ms", r["kernel_ms_per_step"])
print(r["by_family_ms_per_step"]); print(r["by_family_tflops"])
h=d["roofline_hbm_stage"]; print("hbm stage", h["ms"], h["frac"], h["tflops"])
print(d.get("parity"), d.get("host_to_host",{}).get("ms_per_utterance"), d.get("cpu_baseline",{}).get("value"))
PY
