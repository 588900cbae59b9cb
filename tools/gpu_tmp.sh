cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pairs.py -x -q -k transpose 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step', d['ms_per_step'], 'parity', d['parity']['max_abs_vs_reference_golden'], 'frac', r['frac']); print(r['by_family_ms_per_step'])"
FV_SPLIT_CONVT=0 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('fp32 convT: ms_per_step', d['ms_per_step']); print(r['by_family_ms_per_step'])"
