#!/bin/bash
# usage: tools/bench_sweep.sh "ENV=.. ENV=.." ... ; prints ms_per_step for each env set
for e in "$@"; do
  r=$(env $e python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  frac=%.3f' % (d['ms_per_step'], d['roofline']['frac']))")
  echo "$e => $r"
done
