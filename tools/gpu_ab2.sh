#!/bin/bash
# A/B of convh tile shapes and cost models at the HiFi-GAN light sizes
for nfw in 4 2; do for sk in 2 4 6; do
  echo "== C=64 NFW=$nfw SKEL=$sk"; FV_CONVH_NFW=$nfw FV_CONVH_SKEL=$sk timeout 200 python tools/pair_bench.py 64 0 1 split 2>&1 | grep "pairs\|stage"
done; done
for sk in 2 4 6; do echo "== C=128 SKEL=$sk"; FV_CONVH_SKEL=$sk timeout 200 python tools/pair_bench.py 128 0 1 split 2>&1 | grep "pairs\|stage"; done
for nfw in 4 2; do echo "== bench NFW=$nfw"; FV_CONVH_NFW=$nfw timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
