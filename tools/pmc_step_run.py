#!/usr/bin/env python
"""Workload for the per-step HBM-traffic counters (tools/pmc_step.sh runs it under `rocprofv3 --pmc ...`): ONE calibration launch with a
known byte count (tpgsr_add over three 1 GiB buffers: 2 GiB read, 1 GiB written, far beyond the 256 MB Infinity Cache), then a few
C3 (or --config) training steps.  tools/pmc_step_report.py turns the counter CSV into bytes per step / per kernel class."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    import bench
    from tpgsr_amd import kernels as K
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n = 1 << 28                                   # 1 GiB of fp32 per buffer
    a, b, c = (torch.ones(n, device=dev) for _ in range(3))
    torch.cuda.synchronize()
    K.add(a, b, n, c)                             # the FIRST add_kernel dispatch of the process = the calibration launch
    torch.cuda.synchronize()
    del a, b, c
    ts, nets = bench.build_step(args.config, dev)
    lr, hr = bench.synthetic_batch(bench.CONFIGS[args.config]["batch"], 1234, dev)
    for _ in range(args.steps):
        ts.step(lr, hr)
    torch.cuda.synchronize()
    print("pmc_step_run: done", flush=True)


if __name__ == "__main__":
    main()
