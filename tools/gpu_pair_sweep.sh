cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
FV_PAIR_W16=1 timeout 600 python -m pytest tests/test_gpu_pairs.py -x -q 2>&1 | tail -3
for w in 0 1; do echo "== w16=$w"; FV_PAIR_STATIC=1 FV_PAIR_W16=$w python tools/pair_bench.py 16 | grep "static\|mrf"; done
FV_PAIR_STATIC=1 FV_PAIR_W16=1 python tools/pair_bench.py 16 240000 4 | grep "static\|mrf"
} > gpurun_out/r2_pair_w16.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_pair_w16.log
