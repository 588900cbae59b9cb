cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for cap in 1024 768 512 384 256; do
FV_GRID_CAP=$cap python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('cap=$cap', 'ms', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms_per_step'],4), {k: round(v,4) for k,v in r['by_family_ms_per_step'].items()})"
done
