cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export FV_HIPCC_FLAGS=-DFV_PAIR_TRACE
python -c "
from fastvocoder_amd import _native
_native.build()" > gpurun_out/convh_trace_build.log 2>&1
{ echo "=== C=64"; timeout 120 python tools/convh_trace.py 64; echo "=== C=128"; timeout 120 python tools/convh_trace.py 128; } 2>&1 | grep -v amdgpu.ids
