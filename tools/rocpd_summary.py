"""Per-kernel totals of a rocprofv3 (ROCm 7) run that wrote a rocpd sqlite database instead of csv files.

  python tools/rocpd_summary.py gpurun_out/hyd_prof/hyd_results.db [--top 25]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':88s} {'calls':>7s} {'total ms':>10s} {'avg us':>9s} {'min us':>9s} {'max us':>9s} {'%':>6s}")
    for r in rows[:top]:
        print(f"{r[0][:88]:88s} {r[1]:7d} {r[2] / 1e3:10.3f} {r[3]:9.2f} {r[4]:9.2f} {r[5]:9.2f} {100 * r[2] / tot:6.1f}")


if __name__ == "__main__":
    main()
