"""Summarise a rocprofv3 rocpd sqlite database (kernel trace and/or PMC counters) as text.
    python scripts/rocpd_summary.py <results.db> [more.db ...]
"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        con = sqlite3.connect(path)
        cur = con.cursor()
        print(f"## {path}")
        rows = list(cur.execute(
            "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
            "from kernels group by name order by sum(duration) desc"))
        tot = sum(r[2] for r in rows) or 1
        print("| kernel | calls | total_ms | avg_ms | min_ms | max_ms | % |")
        print("|---|---|---|---|---|---|---|")
        for n, c, s, a, mn, mx in rows:
            print(f"| {n[:80]} | {c} | {s/1e6:.3f} | {a/1e6:.4f} | {mn/1e6:.4f} | {mx/1e6:.4f} | {100*s/tot:.1f} |")
        try:
            cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
            if cols:
                q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                     "group by kernel_name, counter_name order by kernel_name, counter_name")
                last = None
                for k, cn, v, n in cur.execute(q):
                    if k != last:
                        print(f"\n### counters: {k[:90]}")
                        last = k
                    print(f"  {cn:32s} avg/dispatch = {v:.6g}   (n={n})")
        except Exception as e:  # no counters in this db
            print("(no counters)", e)
        print()


if __name__ == "__main__":
    main()
