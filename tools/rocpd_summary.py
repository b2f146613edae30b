#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration.
usage: rocpd_summary.py results.db [out.md] [--skip-first-frac F]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    out = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {a[0]} | {a[1] / 1e3:.3f} | {a[1] / a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100 * a[1] / total:.1f} |")
    span = (rows[-1][2] - rows[0][1]) / 1e6 if rows else 0
    lines.append("")
    lines.append(f"total kernel time {total / 1e3:.3f} ms over {len(rows)} dispatches; first-to-last span {span:.3f} ms")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
