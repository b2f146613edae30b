#!/usr/bin/env python
"""Un-profiled phase times of the C3 step on the MAIN stream (HIP events at the phase boundaries, TPGSRTrainStep._marks): where the
caller's stream spends the step, including what it waits for (teacher, SR prologue, side / leaf streams)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from tpgsr_amd import kernels as K  # noqa: E402

K.set_conv_prec(os.environ.get("TPGSR_CONV_PREC", "x2"))
dev = torch.device("cuda", 0)
ts, nets = bench.build_step("c3", dev)
lr, hr = bench.synthetic_batch(48, 1234, dev)
for _ in range(15):
    ts.step(lr, hr)
torch.cuda.synchronize()
acc, n = {}, 0
for rep in range(30):
    ts._marks = []
    ts.step(lr, hr)
    torch.cuda.synchronize()
    prev = ts._marks[0][1]
    for name, ev in ts._marks[1:]:
        acc[name] = acc.get(name, 0.0) + prev.elapsed_time(ev)
        prev = ev
    n += 1
ts._marks = None
tot = sum(acc.values())
print("| phase (main stream, un-profiled, mean of 30 isolated steps) | ms |\n|---|---|")
for k, v in acc.items():
    print(f"| {k} | {v / n:.3f} |")
print(f"| total | {tot / n:.3f} |")
