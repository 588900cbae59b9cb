import glob, sys
files = sorted(glob.glob("gpurun_out/sw_*.log"), key=lambda f: int(f.split("_")[-1].split(".")[0]))
cols = []
names = None
for f in files:
    rows = [l.split() for l in open(f) if l.startswith(("conv", "res", "convT"))]
    if names is None:
        names = [" ".join(r[:2]) + " " + " ".join(r[2:7]) if r[0] in ("res", "convT") else " ".join(r[:1]) + " " + " ".join(r[1:6]) for r in rows]
    cols.append([float(r[7] if r[0] in ("res", "convT") else r[6]) for r in rows])
    print(f"col {len(cols)-1}: {open(f.replace('.log', '.env')).read().strip()}")
for i, n in enumerate(names):
    vals = [c[i] for c in cols]
    best = min(vals)
    print(f"{n:38s} " + " ".join(f"{v:7.1f}{'*' if v == best else ' '}" for v in vals))
print("sum".ljust(38), " ".join(f"{sum(c):8.1f}" for c in cols))
