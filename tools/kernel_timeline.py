#!/usr/bin/env python3
"""Per-kernel duration AND the idle gap in front of each launch, from a rocprofv3 --kernel-trace rocpd database: what a chain of
short dependent kernels (the GPT decode step) really costs.   python tools/kernel_timeline.py <db> [name-filter] [last_n]"""
import collections
import sqlite3
import sys


def main(db, flt="", last_n=0):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    if last_n:
        rows = rows[-last_n:]
    st = collections.OrderedDict()
    prev_end = None
    for name, s, e in rows:
        key = name.split("(")[0][-70:]
        d = st.setdefault(key, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
        if prev_end is not None:
            d[2] += max(0.0, (s - prev_end) / 1e3)
        prev_end = max(prev_end or e, e)
    tot_d = sum(v[1] for v in st.values())
    tot_g = sum(v[2] for v in st.values())
    print(f"{len(rows)} dispatches, span {(rows[-1][2] - rows[0][1]) / 1e6:.2f} ms, kernel time {tot_d / 1e3:.2f} ms, gaps {tot_g / 1e3:.2f} ms")
    print(f"{'calls':>7} {'avg_us':>8} {'gap_us':>8} {'tot_ms':>9}  kernel")
    for k, (n, d, g) in sorted(st.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        if flt and flt not in k:
            continue
        print(f"{n:7d} {d / n:8.2f} {g / n:8.2f} {(d + g) / 1e3:9.2f}  {k}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", int(sys.argv[3]) if len(sys.argv) > 3 else 0)
