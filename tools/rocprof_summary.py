#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) as a per-kernel table (profiles/*.txt)."""
import sqlite3
import sys


def main(db_path, out_path=None, title=""):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                            "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    lines = [title, f"source: {db_path}", f"total kernel time {tot:.2f} ms over {sum(r[1] for r in rows)} dispatches", "",
             f"{'total_ms':>10} {'pct':>6} {'calls':>7} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'vgpr':>5} {'agpr':>5} {'lds':>7}  kernel"]
    for r in rows:
        lines.append(f"{r[2]:10.2f} {100*r[2]/tot:6.2f} {r[1]:7d} {r[3]:10.2f} {r[4]:9.2f} {r[5]:10.2f} {r[6] or 0:5d} {r[7] or 0:5d} {r[8] or 0:7d}  {r[0][:120]}")
    txt = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else "")
