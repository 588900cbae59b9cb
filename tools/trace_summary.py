"""Summarise a rocprofv3 kernel-trace database (rocpd sqlite, *_results.db):
per-kernel launch count, total / average / min / max duration."""
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                     "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':100s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for r in rows[:top]:
        print(f"{r[0][:100]:100s} {r[1]:6d} {r[2]:11.1f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f} {100*r[2]/tot:6.1f}")
    print(f"total kernel time {tot/1e3:.3f} ms over {sum(r[1] for r in rows)} launches")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
