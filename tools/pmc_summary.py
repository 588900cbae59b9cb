"""Per-kernel PMC counter sums from a rocprofv3 --pmc run (rocpd sqlite)."""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
rows = c.execute("select * from counters_collection").fetchall()
ix = {n: i for i, n in enumerate(cols)}
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for r in rows:
    name = r[ix["kernel_name"]] if "kernel_name" in ix else r[ix["name"]]
    agg[name][r[ix["counter_name"]]] += r[ix["value"]]
    cnt[(name, r[ix["counter_name"]])] += 1
for name, d in agg.items():
    if "conv" not in name and len(sys.argv) < 3:
        continue
    print(name[:110])
    for k, v in sorted(d.items()):
        n = cnt[(name, k)]
        print(f"   {k:32s} total={v:16.0f}  per-dispatch={v / n:14.0f}  (n={n})")
