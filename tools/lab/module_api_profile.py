#!/usr/bin/env python
"""Where does the host time of the DROP-IN path go?  bench.py's `module_api` leg (the reference's loop body on the nn.Module API: autograd,
clip_grad_norm_, torch.optim.Adam) under cProfile, 20 steps at bs 48; prints the top functions by cumulative and by own time and the step's
time with the pieces switched one at a time (fused=True Adam, foreach clip)."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tpgsr_amd import kernels as K  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    K.set_conv_prec("x2")
    bench.K_POLICY = K.POLICY
    lr, hr = bench.synthetic_batch(48, 1234, dev)
    r = bench.module_api_bench("c3", dev, lr, hr, steps=20, warmup=6)
    print("module_api:", r["ms_per_step"], "ms/step, host submission", r["host_submission_ms_per_step"])
    pr = cProfile.Profile()
    pr.enable()
    r = bench.module_api_bench("c3", dev, lr, hr, steps=20, warmup=6)
    pr.disable()
    for key in ("cumulative", "tottime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
        print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()
