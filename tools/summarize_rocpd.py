#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace run (rocpd sqlite database) as a markdown table.
    python tools/summarize_rocpd.py RESULTS.db [title] > profiles/rNN_xxx.md"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    rows = db.execute("select name, count(*), avg(end - start), min(end - start), sum(end - start) from kernels group by name order by 5 desc").fetchall()
    total = sum(r[4] for r in rows)
    print(f"# {title}\n")
    print(f"{len(rows)} distinct kernels, {sum(r[1] for r in rows)} launches, {total / 1e6:.2f} ms of kernel time in total.\n")
    print("| kernel | launches | avg µs | min µs | share |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, avg, mn, tot in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
        short = name.replace("(anonymous namespace)::", "").split("(")[0][:80]
        print(f"| `{short}` | {calls} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {100 * tot / total:.1f} % |")


if __name__ == "__main__":
    main()
