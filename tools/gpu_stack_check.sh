#!/bin/bash
# MelGAN / Basis-MelGAN ResidualStack convs on the split-f16 conv kernels: operator + model parity, then configs 1 and 4
# with the stack convs on the fp32 kernels (FV_SPLIT_STACK=0) and on the split-f16 kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pairs.py -x -q -k "reflection or conv1d_split" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
for v in 0 1; do
  echo "=== FV_SPLIT_STACK=$v"
  for i in 0 3; do FV_SPLIT_STACK=$v timeout 300 python tools/bench_configs.py --only $i --steps 10 2>&1 | grep -v amdgpu.ids; done
done
