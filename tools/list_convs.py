#!/usr/bin/env python
"""CPU (TPGSR_PLAN_DRYRUN=1): every MFMA conv launch of one C3 training step, grouped by shape, with its algorithmic GFLOP and the
kernel family the launcher will pick (halo kernel eligibility restated from conv_halo_xbf_launch / wgrad_halo_shape_ok).
usage: TPGSR_PLAN_DRYRUN=1 python tools/list_convs.py [batch]"""
import collections
import os
import sys

os.environ.setdefault("TPGSR_PLAN_DRYRUN", "1")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import tpgsr_oracle as O  # noqa: E402  (tool, not product: synthetic batch only)
from tpgsr_amd import kernels as K  # noqa: E402
from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep  # noqa: E402
from tpgsr_amd.model import tsrn  # noqa: E402
from tpgsr_amd.model.crnn import crnn  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
lr, hr = O.synthetic_batch(N, 1)
sr = tsrn.TSRN_TL(STN=True, mask=True).train()
teacher = crnn.CRNN(32, 1, 37, 256).eval()
stu = crnn.CRNN(32, 1, 37, 256).train()
ts = TPGSRTrainStep([sr], [stu], teacher, stu_iter=1)
ts.step(lr, hr)


def halo_ok(c):
    taps = c.KH * c.KW
    Wp = c.W + 2 * c.pad_w
    return taps >= 2 and c.Cin % 32 == 0 and c.stride_w <= 1 and c.in_dil_w <= 1 and Wp >= 8 and not c.in_b


rows = collections.OrderedDict()
for who, net in (("sr", sr), ("student", stu), ("teacher", teacher)):
    for pl in net._engine()._plans.values():
        for pname, plan in pl.items():
            if pname == "ws":
                continue
            for name, fn, args, sid in plan.ops:
                if name not in ("tpgsr_conv_fwd", "tpgsr_conv_wgrad"):
                    continue
                a = args[0]._obj
                c = a.c if name == "tpgsr_conv_wgrad" else a
                M = c.N * c.OH * c.OW
                Kd = c.KH * c.KW * c.Cin
                key = (who, pname, name[11:], c.N, c.H, c.W, c.Cin, c.Cout, c.KH, c.KW, (c.in_act, c.in_ps, int(bool(c.in2)), int(bool(c.in_scale))), halo_ok(c))
                r = rows.setdefault(key, [0, 0.0])
                r[0] += 1
                r[1] += 2.0 * M * Kd * c.Cout / 1e9
print(f"{'net':8s} {'plan':4s} {'op':6s} {'N':>3s} {'H':>3s} {'W':>4s} {'Cin':>4s} {'Cout':>4s} {'k':>5s} ld halo  n   GFLOP  us@417TF")
tot = collections.Counter()
for k, (n, gf) in rows.items():
    who, pname, op, n_, H, W, Ci, Co, KH, KW, ld, halo = k
    print(f"{who:8s} {pname:4s} {op:6s} {n_:3d} {H:3d} {W:4d} {Ci:4d} {Co:4d} {KH}x{KW:<3d} {str(ld):14s} {'Y' if halo else '-':4s} {n:2d} {gf:7.2f} {gf / 416.7 * 1e3 / n:8.1f}")
    tot[(op, halo)] += gf
print({f"{k[0]}:{'halo' if k[1] else 'tile'}": round(v, 1) for k, v in tot.items()})

# every launch of the step by entry point (plans only; the train-step driver adds its own loss / prior / optimiser launches)
names = collections.Counter()
for who, net in (("sr", sr), ("student", stu), ("teacher", teacher)):
    for pl in net._engine()._plans.values():
        for pname, plan in pl.items():
            if pname == "ws" or (who == "student" and pname == "dgray"):
                continue
            for name, fn, args, sid in plan.ops:
                names[(who, pname, name, sid)] += 1
print()
for (who, pname, name, sid), n in sorted(names.items()):
    print(f"{who:8s} {pname:4s} q{sid} {name:36s} {n}")
