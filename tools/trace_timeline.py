#!/usr/bin/env python
"""Two-stream timeline of ONE training step from a rocprofv3 kernel trace (rocpd sqlite): per-queue busy time, overlap,
idle gaps and the per-kernel breakdown of each queue.  usage: trace_timeline.py results.db [step_index_from_end]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = db.execute("select name, queue_id, start, end from kernels order by start").fetchall()
    # a step starts after the last optimiser launch of the previous one (C3 has several pack programs per step)
    starts = [i + 1 for i, r in enumerate(rows[:-1]) if r[0].startswith("adam_step") and not rows[i + 1][0].startswith("adam_step")]
    a, b = starts[-back - 1], starts[-back]
    step = rows[a:b]
    t0 = step[0][2]
    span = (max(r[3] for r in step) - t0) / 1e3
    print(f"step: {len(step)} dispatches, span {span:.1f} us")
    by_q = defaultdict(list)
    for n, q, s, e in step:
        by_q[q].append((n, (s - t0) / 1e3, (e - t0) / 1e3))
    for q, ks in by_q.items():
        busy = sum(e - s for _, s, e in ks)
        print(f"queue {q}: {len(ks)} kernels, busy {busy:.1f} us, first {ks[0][1]:.1f} last {ks[-1][2]:.1f}")
    # union / overlap
    ev = []
    for n, q, s, e in step:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, t_by_depth = 0, ev[0][0], defaultdict(float)
    for t, d in ev:
        t_by_depth[depth] += (t - last) / 1e3
        depth += d
        last = t
    print("time with N kernels in flight:", {k: round(v, 1) for k, v in sorted(t_by_depth.items())})
    # the longest stretches with NOTHING in flight: what ended before, what started after
    iv = sorted((s, e, n) for n, q, s, e in step)
    gaps, cur_end, last_name = [], iv[0][1], iv[0][2]
    for s_, e_, n_ in iv[1:]:
        if s_ > cur_end:
            gaps.append(((s_ - cur_end) / 1e3, (cur_end - t0) / 1e3, last_name[:40], n_[:40]))
        if e_ > cur_end:
            cur_end, last_name = e_, n_
    print("idle gaps (us, at us, after, before):")
    for g in sorted(gaps, reverse=True)[:12]:
        print(f"   {g[0]:7.1f} at {g[1]:8.1f}   {g[2]}  ->  {g[3]}")
    print(f"   {len(gaps)} gaps, total {sum(g[0] for g in gaps):.1f} us; of them > 10 us: {sum(g[0] for g in gaps if g[0] > 10):.1f} us")
    # phases: find first bwd kernel (tail_bwd) and optimizer (sumsq)
    for mark in ("tail_bwd_kernel", "sumsq", "adam"):
        for n, q, s, e in step:
            if mark in n:
                print(f"first {mark}: {(s - t0) / 1e3:.1f} us")
                break
    for q, ks in by_q.items():
        agg = defaultdict(lambda: [0, 0.0])
        for n, s, e in ks:
            k = n.split("(")[0][:40]
            agg[k][0] += 1
            agg[k][1] += e - s
        print(f"--- queue {q}")
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"   {k:42s} {c:4d} {t:8.1f} us")
    if "--dump" in sys.argv:
        for n, q, s, e in step:
            print(f"{q} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {n[:60]}")


if __name__ == "__main__":
    main()
