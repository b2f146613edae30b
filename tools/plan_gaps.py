#!/usr/bin/env python
"""Un-profiled per-op time line of a train step ("find the missing ~1.1 ms").

The plan executor's stamp mode (tpgsr_plan_set_stamp, include/tpgsr_hip.h) records a timing event behind every launch of every
recorded plan, on the stream the launch runs on; all events are read against one origin recorded at the start of the step.  The step
is measured twice on the same recorded plans: under the default three-stream schedule and under the SERIAL schedule (one stream,
recording order: an op's slot there = its kernel time + one launch boundary, nothing runs next to it).  Per op:

    slot3  = end(op) - end(previous op on the same stream)          three streams
    slot1  = the same under the serial schedule
    extra  = slot3 - slot1  (> 0: it waited for another stream, or shared the machine with one; < 0: cannot happen beyond noise)

and per stream the busy sum, so `wall - sum(slot1 of the main stream's ops)` is split into cross-stream waits and contention, op by op.

    python tools/plan_gaps.py [--config c3] [--prec x2] [--out profiles/r04_plan_gaps_c3_x2.md]
"""
import argparse
import ctypes as C
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def all_plans(nets):
    out = []
    for net in nets:
        eng = net._engine()
        role = getattr(eng, "role", "")
        for j, (key, pl) in enumerate(eng._plans.items()):
            for pname in ("pack_late", "pre", "fwd", "fwd_b", "bwd", "bwd_b", "dgray"):
                if pname in pl and pl[pname]._native:
                    out.append((f"{type(net).__name__}{'(' + role + ')' if role else ''}.{pname}#{j}", pl[pname]))
    return out


def read(plans, lib):
    rows = []
    for pname, pl in plans:
        n = len(pl.ops)
        ms = (C.c_float * n)()
        sid = (C.c_longlong * n)()
        got = lib.tpgsr_plan_read_stamps(pl._native, ms, sid, n)
        if got < 0:
            continue       # this plan did not run in the last step
        for i in range(got):
            if ms[i] >= 0:
                rows.append(dict(plan=pname, op=i, name=pl.ops[i][0].replace("tpgsr_", ""), sid=sid[i], end=1e3 * ms[i]))
    return rows


def slots(rows, serial):
    """slot = end - end of the previous op on the same stream (serial: of the previous op at all)"""
    rows = sorted(rows, key=lambda r: r["end"])
    last = {}
    for r in rows:
        k = 0 if serial else r["sid"]
        r["slot"] = r["end"] - last.get(k, 0.0)
        last[k] = r["end"]
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--prec", default="x2")
    ap.add_argument("--out", default=None)
    ap.add_argument("--steps", type=int, default=9, help="measurements per schedule (each = the last of 6 back-to-back steps)")
    args = ap.parse_args()
    import bench
    from tpgsr_amd import _lib, kernels as K
    lib = _lib.load()
    K.set_conv_prec(args.prec)
    dev = torch.device("cuda", 0)
    ts, nets = bench.build_step(args.config, dev)
    B = bench.CONFIGS[args.config]["batch"]
    lr, hr = bench.synthetic_batch(B, 1234, dev)
    for _ in range(10):
        ts.step(lr, hr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ts.step(lr, hr)
    torch.cuda.synchronize()
    wall_plain = (time.perf_counter() - t0) / 20 * 1e3

    def stamped(serial):
        """every measurement = the LAST of 6 back-to-back steps (the host runs ahead of the GPU as in training: no start-up bubble);
        the origin is recorded on the caller's stream right before that step, i.e. it fires when the previous step's main chain ends"""
        K.set_schedule(serial=serial)
        lib.tpgsr_plan_set_stamp(1)
        acc = None
        walls = []
        for it in range(args.steps):
            torch.cuda.synchronize()
            for _ in range(5):
                ts.step(lr, hr)
            lib.tpgsr_plan_stamp_epoch(torch.cuda.current_stream().cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ts.step(lr, hr)
            e1.record()
            torch.cuda.synchronize()
            if it < 1:
                continue
            walls.append(e0.elapsed_time(e1))
            rows = slots(read(all_plans(nets), lib), serial)
            if acc is None:
                acc = {(r["plan"], r["op"]): dict(r, n=1) for r in rows}
            else:
                for r in rows:
                    a = acc[(r["plan"], r["op"])]
                    a["slot"] += r["slot"]
                    a["end"] += r["end"]
                    a["n"] += 1
        lib.tpgsr_plan_set_stamp(0)
        K.set_schedule()
        for a in acc.values():
            a["slot"] /= a["n"]
            a["end"] /= a["n"]
        return acc, sum(walls) / len(walls)

    three, wall3 = stamped(False)
    one, wall1 = stamped(True)
    main_h = torch.cuda.current_stream().cuda_stream
    names = {main_h: "main", K.side_stream(dev).cuda_stream: "weight-gradient", K.aux_stream(dev).cuda_stream: "teacher / leaf"}
    lines = [f"# Per-op time line of one {args.config.upper()} train step ({args.prec}, bs {B}), un-profiled: timing events behind every plan launch (tools/plan_gaps.py)",
             "",
             f"wall per step: {wall_plain:.3f} ms plain; {wall3:.3f} ms with the stamp events (three streams); {wall1:.3f} ms serial schedule (one stream, recording order)",
             ""]
    per = {}
    for k, a in three.items():
        s1 = one[k]["slot"] if k in one else float("nan")
        per.setdefault(a["sid"], []).append((a, s1))
    lines += ["| stream | plan launches | sum of slots, three streams (us) | the same ops alone = serial slots (us) | waits + contention (us) |", "|---|---|---|---|---|"]
    for sid in sorted(per, key=lambda h: list(names).index(h) if h in names else 99):
        s3 = sum(a["slot"] for a, _ in per[sid])
        s1 = sum(x for _, x in per[sid])
        lines.append(f"| {names.get(sid, hex(sid))} | {len(per[sid])} | {s3:.0f} | {s1:.0f} | {s3 - s1:.0f} |")
    n_ops = sum(len(p.ops) for _, p in all_plans(nets))
    lines += ["", f"plan ops in the step (launches + stream edges): {n_ops}; launches stamped: {len(three)}.  Launches the step makes outside "
              "recorded plans (losses, softmax / prior, optimiser: ~25) show up as part of the next plan launch's slot.", "",
              "## Main stream, the 40 ops with the largest (three-stream slot - serial slot)", "",
              "| plan | op | kernel | end (us) | slot, three streams | slot, serial | extra |", "|---|---|---|---|---|---|---|"]
    main_rows = sorted(per.get(main_h, []), key=lambda t: -(t[0]["slot"] - t[1]))
    for a, s1 in main_rows[:40]:
        lines.append(f"| {a['plan']} | {a['op']} | {a['name']} | {a['end']:.0f} | {a['slot']:.1f} | {s1:.1f} | {a['slot'] - s1:+.1f} |")
    lines += ["", "## Every stamped launch in end-time order (three streams)", "", "| end (us) | stream | plan | op | kernel | slot | serial slot |", "|---|---|---|---|---|---|---|"]
    for k, a in sorted(three.items(), key=lambda kv: kv[1]["end"]):
        s1 = one[k]["slot"] if k in one else float("nan")
        lines.append(f"| {a['end']:.0f} | {names.get(a['sid'], hex(a['sid']))} | {a['plan']} | {a['op']} | {a['name']} | {a['slot']:.1f} | {s1:.1f} |")
    text = "\n".join(lines) + "\n"
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(text)
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main()
