#!/bin/bash
# usage: tools/sweep.sh NAME "ENV=.. ENV=.." ... ; runs conv_bench per env set into gpurun_out/sw_NAME_i.log
mkdir -p gpurun_out
i=0
for e in "$@"; do
  env $e python tools/conv_bench.py $SWEEP_ARGS > gpurun_out/sw_$i.log 2>&1
  echo "$e" > gpurun_out/sw_$i.env
  i=$((i+1))
done
