cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for t in 0 1 2 3; do echo "== tilt=$t"; FV_PAIR_TILT=$t python tools/pair_bench.py 16 | grep -v alone; done
FV_PAIR_TILT=1 timeout 600 python -m pytest tests/test_gpu_pairs.py -x -q 2>&1 | tail -3
} > gpurun_out/r2_pair_tilt.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_pair_tilt.log
