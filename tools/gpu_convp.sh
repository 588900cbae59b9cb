#!/bin/bash
unset FV_PAIR64_UNFUSED
echo "== fused"; timeout 200 python tools/pair_bench.py 64 0 1 split 2>&1 | grep "pairs\|stage\|alone"
echo "== unfused"; FV_PAIR64_UNFUSED=1 timeout 200 python tools/pair_bench.py 64 0 1 split 2>&1 | grep "pairs\|stage"
for sk in 3 8; do echo "== fused skel $sk"; FV_CONVP_SKEL=$sk timeout 200 python tools/pair_bench.py 64 0 1 split 2>&1 | grep "pairs"; done
echo -n "bench fused "; timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']['by_family_ms_per_step']; print(d['ms_per_step'], d['parity']['max_abs_vs_reference_golden'], r['convh64'], r['convh128'])"
echo -n "bench unfused "; FV_PAIR64_UNFUSED=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']['by_family_ms_per_step']; print(d['ms_per_step'], d['parity']['max_abs_vs_reference_golden'], r['convh64'], r['convh128'])"
