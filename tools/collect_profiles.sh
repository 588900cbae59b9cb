#!/bin/bash
# Round profile collection on the GPU box (run from the repo root through gpurun):
#   tools/collect_profiles.sh r02
# writes gpurun_out/<tag>/...; turn it into the committed profiles/<tag>_* with tools/summarise_profiles.py <tag>
TAG=${1:-r06}
R=$PWD
mkdir -p $R/gpurun_out/$TAG
export TMPDIR=/tmp
export FV_BENCH_MFMA_PEAK=0     # (the measured-peak leg of `roofline` is not part of the profiled passes)
B="python $R/bench.py --no-cpu-baseline --no-job --no-exact --no-others"
# --- the headline bench (BASELINE config 2): kernel stats, HBM traffic (two PMC passes), matrix-pipe counters ---
# (the kernel-stats pass is the bench command as it is -- sustained-rate prewarm included, so its per-kernel averages are the
#  sustained ones the line reports; the counter passes run without the prewarm: counts per dispatch do not depend on the clock)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -o bench -- $B --steps 20 --warmup 3 > $R/gpurun_out/$TAG/bench_stats.log 2>&1
FV_BENCH_PREWARM_S=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/fetch -o bench -- $B --steps 5 --warmup 2 > $R/gpurun_out/$TAG/bench_fetch.log 2>&1
FV_BENCH_PREWARM_S=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/write -o bench -- $B --steps 5 --warmup 2 > $R/gpurun_out/$TAG/bench_write.log 2>&1
FV_BENCH_PREWARM_S=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/mfma -o bench -- $B --steps 5 --warmup 2 > $R/gpurun_out/$TAG/bench_mfma.log 2>&1
FV_BENCH_MFMA_PEAK=1 python $R/bench.py --steps 50 --warmup 5 > $R/gpurun_out/$TAG/bench.json 2>/dev/null
# --- the other BASELINE configs (tools/bench_configs.py indices: 0 MelGAN B=1, 2 MB-light B=32, 3 Basis B=64, 4 HiFi-GAN large B=64) ---
for i in 0 2 3 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/cfg$i -o cfg -- python $R/tools/bench_configs.py --only $i --steps 3 > $R/gpurun_out/$TAG/cfg$i.log 2>&1
done
python $R/tools/bench_configs.py --steps 5 > $R/gpurun_out/$TAG/configs.log 2>&1
# --- matrix-cycle accounting at saturation: HiFi-GAN light B = 16 (index 5) and configs 3, 4, 5 -- executed MFMA
#     instructions (SQ_INSTS_MFMA) next to the algorithmic FLOP of the same forward (the library's measurement hook) ---
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
for i in 5 2 3 4; do
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/sat$i -o sat -- python $R/tools/bench_configs.py --only $i --steps 1 --families-json $R/gpurun_out/$TAG/sat${i}_families.json > $R/gpurun_out/$TAG/sat$i.log 2>&1
done
tail -c 300 $R/gpurun_out/$TAG/bench.json; cat $R/gpurun_out/$TAG/configs.log
