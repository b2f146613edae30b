"""CPU (TPGSR_PLAN_DRYRUN=1): structure of the recorded plans that round 4 changed -- checked on the op lists, nothing is computed:
* the SR network's backward plan reduces the LEAF stream's weight-gradient slabs (STN head) on the leaf stream and everything else on the
  weight-gradient stream (two tpgsr_wgrad_reduce_program launches, stream ids 2 and 1), and no longer orders the leaf stream into the
  weight-gradient stream before the latter's reduce;
* the text-prior generator's forward pass is two plans cut in front of conv3, with the operands behind the cut packed by a plan of their own;
  its two BiLSTM layers hand their weight gradients over as batched launches (5 + 3 + 2 GEMMs);
* a GruBlock's weight gradients are one launch per block."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

SCRIPT = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
from oracle import tpgsr_oracle as O
from tpgsr_amd import kernels as K
assert K.DRYRUN
K.set_conv_prec("x2")
from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
from tpgsr_amd.model import tsrn
from tpgsr_amd.model.crnn import crnn
sr = tsrn.TSRN_TL(STN=True, mask=True).train()
teacher, stu = crnn.CRNN(32, 1, 37, 256).eval(), crnn.CRNN(32, 1, 37, 256).train()
lr, hr = O.synthetic_batch(4, 1)
ts = TPGSRTrainStep([sr], [stu], teacher, stu_iter=1)
ts.step(lr, hr)
out = {}
spl = [pl for pl in sr._engine()._plans.values() if "bwd" in pl and len(pl["bwd"])][0]
ops = spl["bwd"].ops
red = [(i, op[3]) for i, op in enumerate(ops) if op[0] == "tpgsr_wgrad_reduce_program"]
out["sr_reduce_streams"] = [sid for _, sid in red]
first_side_reduce = min(i for i, sid in red if sid == 1)
out["leaf_to_side_edges_before_side_reduce"] = sum(1 for op in ops[:first_side_reduce] if op[0] == "edge" and tuple(op[2]) == (2, 1))
out["sr_gru_wgrad"] = sum(1 for op in ops if op[0] == "tpgsr_gru_wgrad")
cpl = [pl for pl in stu._engine()._plans.values() if "bwd" in pl and len(pl["bwd"])][0]
out["crnn_plans"] = sorted(k for k in cpl if k != "ws")
out["crnn_fwd_a"] = [op[0] for op in cpl["fwd"].ops if op[0] in ("tpgsr_pack_program", "tpgsr_conv_fwd", "tpgsr_im2col3x3_c1")]
out["crnn_fwd_b_convs"] = sum(1 for op in cpl["fwd_b"].ops if op[0] == "tpgsr_conv_fwd")
out["crnn_pack_late"] = [op[0] for op in cpl["pack_late"].ops]
out["crnn_batches"] = sorted(len(v) for p in ("bwd", "bwd_b") if p in cpl for v in cpl[p].meta.values())
print("JSON" + json.dumps(out))
'''


@pytest.mark.timeout(600)
def test_round4_plan_structure_without_gpu():
    env = dict(os.environ, TPGSR_PLAN_DRYRUN="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], capture_output=True, text=True, env=env, timeout=550)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("JSON")][-1][4:])
    assert sorted(res["sr_reduce_streams"]) == [1, 2], res
    assert res["leaf_to_side_edges_before_side_reduce"] == 0, res
    assert res["sr_gru_wgrad"] == 10, res
    assert {"fwd", "fwd_b", "pack_late", "bwd", "bwd_b"} <= set(res["crnn_plans"]), res
    assert res["crnn_fwd_a"][0] == "tpgsr_pack_program" and res["crnn_fwd_a"].count("tpgsr_conv_fwd") == 3, res      # conv0 (as 1x1) .. conv2
    assert res["crnn_fwd_b_convs"] >= 4 + 4 and res["crnn_pack_late"][0] == "tpgsr_pack_program", res
    assert res["crnn_batches"] == [2, 3, 5], res
