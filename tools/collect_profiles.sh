#!/bin/bash
# Round profile collection on the GPU box (run from the repo root through gpurun):
#   tools/collect_profiles.sh r01
# writes gpurun_out/<tag>/{stats,fetch,write,mfma}/...; summarise with tools/summarise_profiles.sh
TAG=${1:-r01}
R=$PWD
mkdir -p $R/gpurun_out/$TAG
export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -o bench -- $B --steps 20 --warmup 3 > $R/gpurun_out/$TAG/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/fetch -o bench -- $B --steps 5 --warmup 2 > $R/gpurun_out/$TAG/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/write -o bench -- $B --steps 5 --warmup 2 > $R/gpurun_out/$TAG/bench_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/mfma -o bench -- $B --steps 5 --warmup 2 > $R/gpurun_out/$TAG/bench_mfma.log 2>&1
python $R/bench.py --steps 50 --warmup 5 > $R/gpurun_out/$TAG/bench.json 2>/dev/null
tail -c 300 $R/gpurun_out/$TAG/bench.json
