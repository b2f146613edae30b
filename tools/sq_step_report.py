#!/usr/bin/env python
"""rocprofv3 SQ-counter CSVs of tools/pmc_step_run.py (tools/sq_step.sh) -> per-kernel table of ONE training step: matrix-pipe busy
fraction, where the waves wait, LDS conflicts, resident waves.
usage: sq_step_report.py OUT_PREFIX csv [csv ...]      (writes OUT_PREFIX.md and OUT_PREFIX.json)
Units (MI355X_MICROARCH.md, "rocprofv3 PMC slots" and the constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles
summed over all waves; SQ_VALU_MFMA_BUSY_CYCLES are cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
  mfma busy       = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8)
  wait / stall / active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint, sum ~ 1):
                    parked at s_waitcnt / barrier, issue stall (MFMA dependency, pipe busy), issuing
  waves / CU      = 4 x SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8) / 256
A step = the dispatches between the last optimiser launches of two consecutive steps; the LAST step is reported.  Counter collection
serialises the dispatches, so every row is the kernel ALONE on the chip (not next to the other two streams)."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:88]


def load(path):
    per = defaultdict(dict)       # dispatch id -> {counter: value}
    names = {}
    for r in csv.DictReader(open(path)):
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        per[d]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])      # duration of the dispatch UNDER counter collection
        names[d] = r["Kernel_Name"]
    ids = sorted(per)
    return [(names[d], per[d]) for d in ids]


def last_step(rows):
    adam = [i for i, (k, _) in enumerate(rows) if k.startswith("adam_step") or "adam_step" in k]
    ends = [i for i in adam if not any(i < j <= i + 20 for j in adam)]
    if len(ends) < 2:
        return rows
    return rows[ends[-2] + 1:ends[-1] + 1]


def main():
    out = sys.argv[1]
    agg = defaultdict(lambda: defaultdict(float))
    count = defaultdict(int)
    for path in sys.argv[2:]:
        step = last_step(load(path))
        seen = defaultdict(int)
        for k, c in step:
            s = short(k)
            seen[s] += 1
            for name, v in c.items():
                # GRBM_GUI_ACTIVE is in both passes: keep the first pass's
                key = name if not (name in ("GRBM_GUI_ACTIVE", "_ns") and path != sys.argv[2]) else name + "_pass2"
                agg[s][key] += v
        for s, n in seen.items():
            count[s] = max(count[s], n)
    rows = []
    for s, c in agg.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        per_xcd = gui / 8.0
        r = dict(kernel=s, launches=count[s], gui_active_per_xcd=per_xcd,
                 us_counter_pass=c.get("_ns", 0.0) / 1e3,
                 mfma_busy=c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * per_xcd) if per_xcd else None,
                 mfma_busy_kernel=c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * c["_ns"] * 2.1) if c.get("_ns") else None,
                 wait_any=c.get("SQ_WAIT_ANY", 0.0) / wc if wc else None,
                 wait_inst_any=c.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else None,
                 wait_inst_lds=c.get("SQ_WAIT_INST_LDS", 0.0) / wc if wc else None,
                 active_inst=c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc if wc else None,
                 waves_per_cu=4.0 * wc / per_xcd / 256.0 if per_xcd else None,
                 lds_conflict=c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else None,
                 insts_valu=c.get("SQ_INSTS_VALU"), insts_lds=c.get("SQ_INSTS_LDS"), insts_vmem=c.get("SQ_INSTS_VMEM"), insts_salu=c.get("SQ_INSTS_SALU"),
                 active_vmem=c.get("SQ_ACTIVE_INST_VMEM"), active_lds=c.get("SQ_ACTIVE_INST_LDS"), active_valu=c.get("SQ_ACTIVE_INST_VALU"),
                 raw={k: v for k, v in c.items()})
        rows.append(r)
    rows.sort(key=lambda r: -r["gui_active_per_xcd"])
    json.dump(rows, open(out + ".json", "w"), indent=1)

    def f(x, p=3):
        return "-" if x is None else f"{x:.{p}f}"
    with open(out + ".md", "w") as fh:
        fh.write("# SQ counters per kernel, one training step (every kernel alone on the chip: counter collection serialises dispatches)\n\n")
        fh.write("`mfma busy` = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD); `parked` = SQ_WAIT_ANY (s_waitcnt / barrier), `stall` = SQ_WAIT_INST_ANY "
                 "(issue stall: MFMA dependency / pipe), `lds stall` = SQ_WAIT_INST_LDS, `issuing` = SQ_ACTIVE_INST_ANY, all over SQ_WAVE_CYCLES; `cycles` = GRBM_GUI_ACTIVE per XCD "
                 "summed over the step's launches of that kernel.  Counter collection adds ~10 us of idle window to EVERY dispatch (a 5-us BatchNorm finalize "
                 "reads 15): the fractions of a 30-40 us kernel are diluted by a quarter to a third -- compare kernels with each other, and take absolute "
                 "matrix-pipe utilisation from FLOPs / un-profiled time (bench.py's roofline).\n\n")
        fh.write("| kernel | launches | cycles (k) | us each (under counter collection) | mfma busy | mfma busy of the kernel's own time (2.1 GHz) | parked | stall | lds stall | issuing | waves / CU | lds conflict |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            n = max(r["launches"], 1)
            fh.write(f"| `{r['kernel']}` | {r['launches']} | {r['gui_active_per_xcd'] / 1e3:.1f} | {r['us_counter_pass'] / n:.1f} | {f(r['mfma_busy'])} | {f(r['mfma_busy_kernel'])} | {f(r['wait_any'])} | "
                     f"{f(r['wait_inst_any'])} | {f(r['wait_inst_lds'])} | {f(r['active_inst'])} | {f(r['waves_per_cu'], 1)} | {f(r['lds_conflict'])} |\n")
    print(open(out + ".md").read()[:6000])


if __name__ == "__main__":
    main()
