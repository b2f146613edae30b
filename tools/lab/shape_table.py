#!/usr/bin/env python
"""Full per-shape table of the convolution family of one C3 step (bench.conv_roofline), for the current environment switches."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from tpgsr_amd import kernels as K  # noqa: E402

K.set_conv_prec(os.environ.get("TPGSR_CONV_PREC", "x2"))
dev = torch.device("cuda", 0)
ts, nets = bench.build_step(sys.argv[1] if len(sys.argv) > 1 else "c3", dev)
lr, hr = bench.synthetic_batch(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]["batch"], 1234, dev)
for _ in range(3):
    ts.step(lr, hr)
torch.cuda.synchronize()
r = bench.conv_roofline(nets, reps=8)
print(json.dumps(dict(total={k: v for k, v in r["total"].items()}, by_kind=r["by_kind"], table=r["table"])))
