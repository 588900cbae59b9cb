#!/bin/bash
# split-f16 transposed conv: operator parity, model parity, the headline step with the upsamplers on the fp32 kernels
# (FV_SPLIT_CONVT=0) and on convt_kernel, the other configs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pairs.py -x -q -k "transpose" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
for v in 0 1; do
  echo "=== FV_SPLIT_CONVT=$v"
  FV_SPLIT_CONVT=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step', d['ms_per_step'], 'parity', d['parity']['max_abs_vs_reference_golden'], 'frac', r['frac']); print(r['by_family_ms_per_step'])"
  for i in 0 2 3 4; do FV_SPLIT_CONVT=$v timeout 300 python tools/bench_configs.py --only $i --steps 5 2>&1 | grep -v amdgpu.ids; done
done
