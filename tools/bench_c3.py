#!/usr/bin/env python
"""C3 (BASELINE config 3) timing: TPGSR-TSRN_TL + CRNN teacher/student text prior, stu_iter 1, bs 48, fp32, one MI355X.
Not the driver's bench line (bench.py measures C2, the configuration the metric is quoted on); prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import tpgsr_oracle as O  # noqa: E402  (weights by recipe only)
from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep  # noqa: E402
from tpgsr_amd.model import tsrn  # noqa: E402
from tpgsr_amd.model.crnn import crnn  # noqa: E402
from bench import synthetic_batch, BATCH  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda", 0)
    sr = tsrn.TSRN_TL(STN=True, mask=True)
    sr.load_state_dict(O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 11, tps_hw=(16, 64)))
    teacher = crnn.CRNN(32, 1, 37, 256)
    teacher.load_state_dict(O.recipe_state_dict(O.crnn_spec(), 12))
    student = crnn.CRNN(32, 1, 37, 256)
    student.load_state_dict(O.recipe_state_dict(O.crnn_spec(), 13))
    ts = TPGSRTrainStep([sr.to(dev).train()], [student.to(dev).train()], teacher.to(dev).eval(), stu_iter=1)
    lr, hr = synthetic_batch(BATCH, 1234, dev)
    for _ in range(5):
        loss = ts.step(lr, hr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = ts.step(lr, hr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_launch = sum(len(p) for m in (sr, student, teacher) for pl in m._engine()._plans.values() for p in (pl["fwd"], pl["bwd"]))
    print(json.dumps({"workload": "C3: TSRN_TL (STN+mask) + CRNN teacher/student prior, stu_iter 1, fp32 train step",
                      "batch": BATCH, "steps": steps, "ms_per_step": round(1e3 * dt / steps, 3),
                      "img_per_s": round(BATCH * steps / dt, 1), "recorded_ops_all_plans": n_launch,
                      "final_loss": float(loss.item())}))


if __name__ == "__main__":
    main()
