set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pairs.py -x -q 2>&1 | tail -30 > gpurun_out/r2_pairs.log
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
FV_PAIR=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench1_nopair.json 2>> gpurun_out/r2_bench1.err
cat gpurun_out/r2_pairs.log gpurun_out/r2_gpu.log
python - <<'PY'
import json
for f in ("gpurun_out/r2_bench1.json","gpurun_out/r2_bench1_nopair.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["roofline"]["frac_whole_step"], d["roofline_hbm_stage"])
    except Exception as e: print(f, "ERR", e)
PY
