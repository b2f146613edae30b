"""CPU: host logic added in round 5 -- the train steps' default arithmetic policy, the policy context, plan caches per policy."""
import os

import pytest

from tpgsr_amd import kernels as K


def test_train_step_policy_precedence(monkeypatch):
    # nobody chose: the benchmarked, gated policy
    monkeypatch.setattr(K, "_POLICY_EXPLICIT", False)
    assert K.train_step_policy(None) == "x2"
    # the step's own argument wins over everything
    assert K.train_step_policy("x3") == "x3"
    with pytest.raises(ValueError):
        K.train_step_policy("fp8")
    # an explicit choice (TPGSR_CONV_PREC / set_conv_prec) is honoured
    prev = K.POLICY
    try:
        K.set_conv_prec("x3b2")
        assert K.train_step_policy(None) == "x3b2"
    finally:
        K.set_conv_prec(prev)
    with pytest.raises(ValueError):
        K.set_conv_prec("nope")


def test_policy_context_restores_and_is_not_an_explicit_choice(monkeypatch):
    monkeypatch.setattr(K, "_POLICY_EXPLICIT", False)
    prev = (K.POLICY, K.CONV_TERMS)
    with K.policy("x2"):
        assert K.POLICY == "x2" and K.CONV_TERMS == 2
        with K.policy("f32"):
            assert K.CONV_TERMS == 0
        assert K.POLICY == "x2"
    assert (K.POLICY, K.CONV_TERMS) == prev and not K._POLICY_EXPLICIT


def test_package_import_asks_for_eight_hardware_queues():
    import tpgsr_amd  # noqa: F401
    assert os.environ.get("GPU_MAX_HW_QUEUES")            # set by the package unless the caller chose a value


_SCRIPT = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
from oracle import tpgsr_oracle as O
from tpgsr_amd import kernels as K
assert K.DRYRUN
from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
from tpgsr_amd.model import tsrn
from tpgsr_amd.model.crnn import crnn
sr = tsrn.TSRN_TL(STN=True, mask=True).train()
teacher, stu = crnn.CRNN(32, 1, 37, 256).eval(), crnn.CRNN(32, 1, 37, 256).train()
lr, hr = O.synthetic_batch(4, 1)
ts = TPGSRTrainStep([sr], [stu], teacher, stu_iter=1)
ts.step(lr, hr)
out = dict(precision=ts.precision)
pl = [p for p in sr._engine()._plans.values() if "bwd" in p and len(p["bwd"])][0]
key = [k for k, p in sr._engine()._plans.items() if p is pl][0]
out["plan_key_policy"] = key[-1]
fwd, bwd = [op[0] for op in pl["fwd"].ops], pl["bwd"].ops
out["fused_gru"], out["scan_alone"] = fwd.count("tpgsr_bigru_proj_fwd"), fwd.count("tpgsr_bigru_fwd")
out["fwd_proj_convs"] = sum(1 for op in pl["fwd"].ops if op[0] == "tpgsr_conv_fwd" and op[2][0]._obj.Cout == 192)
out["shiftsum_nhwc_stream"] = [op[3] for op in bwd if op[0] == "tpgsr_shiftsum_nhwc"]
hs = [i for i, op in enumerate(bwd) if op[0] == "tpgsr_hsum"]
out["hsum_streams"] = [bwd[i][3] for i in hs]
joins = [i for i, op in enumerate(bwd) if op[0] == "edge" and tuple(op[2]) == (2, 0)]
first_stn = min(i for i, op in enumerate(bwd) if op[0] == "tpgsr_grid_sample_bwd")
first_ig = min(i for i, op in enumerate(bwd) if op[0] == "tpgsr_strip_resample_bwd")
out["leaf_join_between"] = [max(hs) < j < min(first_stn, first_ig) for j in joins]
print("JSON" + json.dumps(out))
'''


@pytest.mark.timeout(600)
def test_round5_plan_structure_without_gpu():
    """TPGSR_PLAN_DRYRUN=1, no policy chosen: the train step records x2; ten GruBlock forwards are one launch each (no projection convolution,
    no scan launch of its own); block1's data gradient ends in the NHWC shift-sum on the leaf stream; the text strip's five H-sums run on
    the leaf stream and ONE leaf -> caller edge follows the last of them, in front of both the STN head's chain and the InfoGen backward."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, TPGSR_PLAN_DRYRUN="1")
    env.pop("TPGSR_CONV_PREC", None)
    r = subprocess.run([sys.executable, "-c", _SCRIPT % dict(root=root)], capture_output=True, text=True, env=env, timeout=550)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("JSON")][-1][4:])
    assert res["precision"] == "x2" and res["plan_key_policy"] == "x2", res
    assert res["fused_gru"] == 10 and res["scan_alone"] == 0 and res["fwd_proj_convs"] == 0, res
    assert res["shiftsum_nhwc_stream"] == [2], res
    assert res["hsum_streams"] == [2] * 5, res
    assert res["leaf_join_between"] == [True], res
