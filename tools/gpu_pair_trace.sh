cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# rebuild with the time-stamp code compiled in (tuning aid only)
FV_HIPCC_FLAGS=-DFV_PAIR_TRACE python -c "
from fastvocoder_amd import _native
_native.build(force=True)" > gpurun_out/r2_trace_build.log 2>&1
{
echo "=== C=16, 1 block per CU, static"; FV_PAIR_STATIC=1 FV_PAIR_BLOCKS=256 python tools/pair_trace.py 16 | grep -v "tile [3-9]" | head -40
echo "=== C=16, 2 blocks per CU, static"; FV_PAIR_STATIC=1 python tools/pair_trace.py 16 | grep -v "tile [4-9]"
echo "=== C=32"; FV_PAIR_STATIC=1 python tools/pair_trace.py 32 | grep -v "tile [2-9]" | head -30
} > gpurun_out/r2_pair_trace2.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_pair_trace2.log
