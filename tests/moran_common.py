"""shared by the MORAN tests: the fixture, the weights it was generated with (rebuilt from the stored recipe seed)"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, GOLDEN)


def fixture():
    return np.load(os.path.join(GOLDEN, "moran_eval.npz"))


def layout():
    return json.load(open(os.path.join(GOLDEN, "moran_layout.json")))["moran"]


def state_dict(seed: int) -> "OrderedDict[str, torch.Tensor]":
    from make_golden_moran import weights           # (imports no reference code: only main() does)
    template = OrderedDict((k, torch.zeros(shape, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32))
                           for k, shape in layout())
    return weights(template, seed)
