cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export FV_HIPCC_FLAGS=-DFV_PAIR_TRACE
python -c "
from fastvocoder_amd import _native
_native.build()" > gpurun_out/convp_trace_build.log 2>&1
for a in "11,7,3" "3" "11"; do echo "=== $a"; timeout 200 python tools/convp_trace.py "$a" 2>&1 | grep -v amdgpu.ids; done
