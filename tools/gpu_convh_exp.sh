#!/bin/bash
# timing experiments on the convh kernels (library built with FV_HIPCC_FLAGS=-DFV_CONVH_EXP; results are wrong with dbg != 0)
export FV_HIPCC_FLAGS="-DFV_CONVH_EXP"
for d in 0 32 4 36 2 8 1 47; do for C in 64 128; do
  echo -n "dbg=$d "; FV_TUNING=1 FV_PAIR_DBG=$d timeout 200 python tools/pair_bench.py $C 0 1 split 2>&1 | grep "pairs dil=5"
done; done
