mkdir -p gpurun_out/s5
for i in 1 2 3; do
for cfg in "20 5" "50 5" "200 20" "20 200" "20 5"; do
set -- $cfg
python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-job --no-exact --no-others 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('steps',d['steps'],'warmup',d['warmup'],'ms_per_step %.4f'%d['ms_per_step'],'default %.4f'%d['ms_per_step_default_policy'])
"
done
done > gpurun_out/s5/warm_ab.txt 2>&1
cat gpurun_out/s5/warm_ab.txt
