cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export FV_HIPCC_FLAGS=-DFV_PAIR_TRACE
python -c "
from fastvocoder_amd import _native
_native.build()" > gpurun_out/chain_trace_build.log 2>&1
for a in "$@"; do echo "=== $a"; timeout 200 python tools/chain_trace.py "$a" 2>&1 | grep -v amdgpu.ids; done
