"""CPU (TPGSR_PLAN_DRYRUN=1): the `--tpg OPT` recogniser's recorded mode (tpgsr_amd/engine_functional.py) -- the module's forward and its
autograd backward are traced into kernels.Plan objects without a GPU, handed to the native executor (entry points + argument counts), and
the recorded launch census is what the network is: 33 convolutions forward, 33 data / 33 weight gradients backward (one per conv + the
prediction layer... minus the input's), one batched slab reduce, a tpgsr_add per residual block for the forked gradients, and no parameter
gradient left to autograd (every parameter has a sink in the arena)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

SCRIPT = r'''
import collections, json, sys, torch
sys.path.insert(0, %(root)r)
from tpgsr_amd import kernels as K
assert K.DRYRUN
import bench
from tpgsr_amd.model.crnn import model as m
K.set_conv_prec("x2")
net = m.Model(bench.OPT_ARGS).train()
eng = net._engine()
gray = torch.rand(4, 1, 32, 100)
logits = eng.forward(gray, True)
dg = eng.backward(4, gray, torch.randn(4, 26, 37), need_dgray=True)
pl = list(eng._plans.values())[0]
out = dict(logits=list(logits.shape), dgray=list(dg.shape), fwd=dict(collections.Counter(op[0] for op in pl["fwd"].ops)),
           bwd=dict(collections.Counter(op[0] for op in pl["bwd"].ops)), native=[bool(pl["fwd"]._native), bool(pl["bwd"]._native)],
           grads_attached=all(p.grad is not None and p.grad.data_ptr() != 0 for p in net.parameters()))
# the eval-mode (teacher) plan and a second batch size are separate traces
net2 = m.Model(bench.OPT_ARGS).eval()
e2 = net2._engine()
out["eval"] = [list(e2.forward(gray, False).shape), list(e2.forward(torch.rand(2, 1, 32, 100), False).shape), len(e2._plans)]
print("JSON" + json.dumps(out))
'''


@pytest.mark.timeout(600)
def test_opt_recogniser_traces_into_plans_without_gpu():
    env = dict(os.environ, TPGSR_PLAN_DRYRUN="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], capture_output=True, text=True, env=env, timeout=550)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("JSON")][-1][4:])
    assert res["logits"] == [4, 26, 37] and res["dgray"] == [4, 1, 32, 100] and res["native"] == [True, True] and res["grads_attached"]
    f, b = res["fwd"], res["bwd"]
    assert f["tpgsr_conv_fwd"] == 33 and f["tpgsr_pack_conv_weight"] == 33 and f["tpgsr_bn_finalize"] == 32 and f["tpgsr_add"] == 11
    assert b["tpgsr_conv_wgrad"] == 33 and b["tpgsr_conv_fwd"] == 33 and b["tpgsr_wgrad_reduce_program"] == 1
    assert b["tpgsr_add"] == 11 and b["tpgsr_bn_bwd_finalize"] == 32 and b["join"] == 1
    assert res["eval"] == [[4, 26, 37], [2, 26, 37], 2]
