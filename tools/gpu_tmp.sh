cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pairs.py -x -q 2>&1 | tail -3
for v in 0 1; do
echo "== FV_SCHED=$v"
for C in 128 64; do FV_SCHED=$v timeout 200 python tools/pair_bench.py $C 0 1 split 2>&1 | grep "pairs\|stage"; done
FV_SCHED=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step', d['ms_per_step'], 'parity', d['parity']['max_abs_vs_reference_golden'], 'frac', r['frac']); print(r['by_family_ms_per_step'])"
done
for sw in 0 2 8; do echo "== switch $sw"; FV_SCHED_SWITCH=$sw timeout 200 python tools/pair_bench.py 128 0 1 split 2>&1 | grep "pairs\|stage"; FV_SCHED_SWITCH=$sw timeout 200 python tools/pair_bench.py 64 0 1 split 2>&1 | grep "pairs\|stage";  done
