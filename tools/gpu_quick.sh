#!/bin/bash
# quick check after a kernel change: pair / conv operator parity, per-launch timing at the four stage sizes, the bench line
timeout 900 python -m pytest tests/test_gpu_pairs.py -x -q 2>&1 | tail -4
for C in 128 64 32 16; do timeout 200 python tools/pair_bench.py $C 0 1 split 2>&1 | grep "pairs\|stage"; done
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step', d['ms_per_step'], 'parity', d['parity']['max_abs_vs_reference_golden'], 'frac', r['frac'], 'hbm', d['roofline_hbm_stage']['frac']); print(r['by_family_ms_per_step'])"
