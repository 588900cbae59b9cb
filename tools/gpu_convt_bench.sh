#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python tools/convt_bench.py 1 2>&1 | grep convT
for d in 8 2 1 11; do FV_TUNING=1 FV_PAIR_DBG=$d timeout 200 python tools/convt_bench.py 1 2>&1 | grep convT | head -2; done
FV_CONVH_BLOCKS=128 timeout 200 python tools/convt_bench.py 1 2>&1 | grep convT | sed -n 2p
timeout 200 python tools/convt_bench.py 16 2>&1 | grep convT
