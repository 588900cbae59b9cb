"""Print kernel launches of a rocprofv3 trace db in order (name, duration us, gap to previous)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count from kernels order by start").fetchall()
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
prev = None
for i, r in enumerate(rows[lo:hi]):
    gap = (r[1] - prev) / 1e3 if prev else 0.0
    prev = r[2]
    nm = r[0].replace("void fv::", "").replace("(fv::ConvParams)", "")
    print(f"{lo+i:5d} {nm[:60]:60s} dur={(r[2]-r[1])/1e3:8.1f}us gap={gap:6.1f} grid={r[3]}x{r[4]} wg={r[5]} lds={r[6]} vgpr={r[7]}+{r[8]}")
