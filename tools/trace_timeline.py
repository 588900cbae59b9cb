"""Timeline of the last forward in a rocprofv3 kernel trace: per kernel start offset, duration, stream."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
# last forward = last 78 conv kernels
convs = [r for r in rows if "conv_" in r[0]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 78
last = convs[-n:]
t0 = last[0][1]
busy_until = {}
for r in last:
    nm = r[0].replace("void fv::", "").replace("(fv::ConvParams)", "").replace("conv_mfma_kernel", "mfma").replace("conv_narrow_kernel", "narrow")
    print(f"{(r[1]-t0)/1e3:9.1f} -> {(r[2]-t0)/1e3:9.1f}  dur={(r[2]-r[1])/1e3:7.1f}us  stream={r[3]}  blocks={r[4]//r[5]:5d}  {nm}")
print(f"total span {(last[-1][2]-t0)/1e3:.1f} us; sum of durations {sum(r[2]-r[1] for r in last)/1e3:.1f} us")
