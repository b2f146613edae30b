#!/usr/bin/env python
"""rocprofv3 counter CSVs of tools/pmc_step_run.py -> HBM bytes per training step and per kernel class.
usage: pmc_step_report.py OUT.json FETCH_csv WRITE_csv
Calibration: the first add_kernel dispatch read 2 GiB and wrote 1 GiB (MI355X_MICROARCH.md, HBM section: FETCH_SIZE under-reports wide
streaming reads by 2x on gfx950, WRITE_SIZE is uncalibrated) -- every counter value is scaled by (known bytes / counted) of that launch.
A step = the dispatches between the last optimiser launches of two consecutive steps; the LAST step is reported."""
import csv
import json
import sys
from collections import defaultdict

CLASSES = [("conv fwd/dgrad (halo)", "conv_halo_xbf_kernel"), ("conv fwd/dgrad (halo)", "conv_halo3_xbf_kernel"),
           ("conv fwd/dgrad (tile loop)", "conv_fwd_xbf_kernel"), ("conv fwd/dgrad (tile loop)", "conv_panel_xbf_kernel"),
           ("conv fwd/dgrad (tile loop)", "conv_fwd_kernel"),
           ("conv wgrad", "conv_wgrad"), ("conv wgrad", "gru_wgrad_kernel"), ("dy_split", "dy_split_kernel"), ("wgrad slab reduce", "wgrad_reduce_program"),
           ("BiGRU", "bigru_"), ("BiLSTM", "lstm_"), ("BatchNorm", "bn_"), ("optimiser", "adam_step")]


def load(path, counter):
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def per_step(rows, known_bytes):
    cal = next(v for _, k, v in rows if k.startswith("add_kernel"))
    scale = known_bytes / cal
    # the last optimiser launch of a step: no further adam_step_kernel among the next 20 dispatches (a step has one per network)
    adam = [i for i, (_, k, _) in enumerate(rows) if k.startswith("adam_step")]
    ends = [i for i in adam if not any(i < j <= i + 20 for j in adam)]
    a, b = ends[-2] + 1, ends[-1] + 1
    step = rows[a:b]
    total = sum(v for _, _, v in step) * scale
    by = defaultdict(float)
    for _, k, v in step:
        for name, pat in CLASSES:
            if pat in k:
                by[name] += v * scale
                break
        else:
            by["other"] += v * scale
    return dict(raw_calibration_value=cal, scale_to_bytes=scale, dispatches=len(step), bytes=total,
                by_class={k: round(v) for k, v in sorted(by.items(), key=lambda kv: -kv[1])})


def main():
    out, fetch_csv, write_csv = sys.argv[1:4]
    res = dict(fetch=per_step(load(fetch_csv, "FETCH_SIZE"), 2.0 * (1 << 30)), write=per_step(load(write_csv, "WRITE_SIZE"), 1.0 * (1 << 30)))
    res["hbm_bytes_per_step"] = res["fetch"]["bytes"] + res["write"]["bytes"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
