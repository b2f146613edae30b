"""CPU: host logic changed in round 6 (dry-run plans, no GPU):
  * three-channel networks (mask=False, the reference's `--mask` default) record again: block1's folded data gradient and the tail take any
    channel count;
  * TPGSR_LEAF_EARLY=1 together with the text strip's gradient on the leaf stream: the leaf -> caller edge is recorded BEFORE the early leaf
    section opens, so the InfoGen backward never reads dtemb unordered."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

_MASK_FALSE = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
from oracle import tpgsr_oracle as O
from tpgsr_amd import kernels as K
assert K.DRYRUN
from tpgsr_amd.interfaces.super_resolution import TSRNTrainStep
from tpgsr_amd.model import tsrn
out = {}
for stn in (True, False):
    sr = tsrn.TSRN(STN=stn, mask=False).train()
    lr, hr = O.synthetic_batch(2, 1)
    ts = TSRNTrainStep(sr)
    ts.step(lr[:, :3].contiguous(), hr[:, :3].contiguous())
    pl = [p for p in sr._engine()._plans.values() if "bwd" in p and len(p["bwd"])][0]
    fwd, bwd = [op[0] for op in pl["fwd"].ops], [op[0] for op in pl["bwd"].ops]
    out["stn%%d" %% stn] = dict(tail=fwd.count("tpgsr_tail_shiftsum_tanh"), folded=bwd.count("tpgsr_shiftsum_nhwc"), tail_bwd=bwd.count("tpgsr_tail_bwd"))
print("JSON" + json.dumps(out))
'''

_LEAF_EARLY = r'''
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
pl = [p for p in sr._engine()._plans.values() if "bwd" in p and len(p["bwd"])][0]
bwd = pl["bwd"].ops
hs = [i for i, op in enumerate(bwd) if op[0] == "tpgsr_hsum"]
joins = [i for i, op in enumerate(bwd) if op[0] == "edge" and tuple(op[2]) == (2, 0)]
first_ig = min(i for i, op in enumerate(bwd) if op[0] == "tpgsr_strip_resample_bwd")
out = dict(hsum_streams=[bwd[i][3] for i in hs], joins_after_last_hsum_before_infogen=[j for j in joins if max(hs) < j < first_ig],
           infogen_stream=bwd[first_ig][3], n_leaf_ops_between=sum(1 for op in bwd[max(hs) + 1:first_ig] if op[0].startswith("tpgsr_") and op[3] == 2))
print("JSON" + json.dumps(out))
'''


def _run(script, **env):
    e = dict(os.environ, TPGSR_PLAN_DRYRUN="1", **env)
    e.pop("TPGSR_CONV_PREC", None)
    r = subprocess.run([sys.executable, "-c", script % dict(root=ROOT)], capture_output=True, text=True, env=e, timeout=550)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("JSON")][-1][4:])


@pytest.mark.timeout(600)
def test_three_channel_networks_record_a_train_step():
    """TSRN(mask=False) -- in_planes 3, the reference's default (main.py: `--mask` is a store_true flag; model/tsrn.py:24-26) -- with and
    without the STN: the plans record (round 5's FoldedDgrad asserted KS Ci % 4 == 0 and the shift-sum launcher rejected KS Co = 27)."""
    res = _run(_MASK_FALSE)
    assert res["stn1"] == dict(tail=1, folded=1, tail_bwd=1), res
    assert res["stn0"] == dict(tail=1, folded=0, tail_bwd=1), res


@pytest.mark.timeout(600)
def test_leaf_early_with_the_strip_on_the_leaf_stream_orders_dtemb():
    """TPGSR_LEAF_EARLY=1 + TPGSR_LEAF_STRIP=1 (default): the strip's H-sums run on the leaf stream (2), block 0's part of the backward pass
    follows on the leaf stream as well, and a 2 -> 0 edge sits between the last H-sum and the InfoGen backward (caller's stream)."""
    res = _run(_LEAF_EARLY, TPGSR_LEAF_EARLY="1")
    assert res["hsum_streams"] == [2] * 5, res
    assert res["infogen_stream"] == 0 and res["n_leaf_ops_between"] > 0, res
    assert len(res["joins_after_last_hsum_before_infogen"]) >= 1, res


@pytest.mark.timeout(900)
def test_no_scan_instruction_takes_its_low_half_from_the_odd_register_of_an_lds_pair():
    """Root cause of round 5's unrepeatable GruBlock forward (profiles/r06_gru_proj_root_cause.md): `v_pk_fma_f32 ... op_sel:[0,1,0]` on a
    register pair a ds_read_b128 had just returned read the odd register as zero in lanes 48-63 under three scanning waves per SIMD.  The
    scans build (v, v) from odd components through a v_mov (gru_common.h: gru_dup_odd).  This compiles the two scan sources for gfx950
    (no GPU needed) and checks that NO packed instruction carries an op_sel that selects the high register for the low half."""
    import re
    import tempfile
    from tpgsr_amd import build as B
    csrc = os.path.join(ROOT, "tpgsr_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for src in ("gru.hip", "gru_proj.hip"):
            out = os.path.join(td, src + ".s")
            cmd = [B._hipcc()] + B.FLAGS + ["-x", "hip", "-S", "--cuda-device-only", os.path.join(csrc, src), "-o", out]
            procs.append((src, out, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        for src, out, p in procs:
            log, _ = p.communicate(timeout=850)
            assert p.returncode == 0, log.decode()[-3000:]
            isa = open(out).read()
            packed = re.findall(r"^\s*v_pk_\w+ .*$", isa, flags=re.M)
            assert len(packed) > 100, (src, len(packed))             # the scans ARE packed multiply-adds
            bad = [ln.strip() for ln in packed if re.search(r"op_sel:\[(0|1),1", ln) or re.search(r"op_sel:\[1", ln)]
            assert not bad, f"{src}: {len(bad)} packed instructions read a high register for their low half, e.g. {bad[:3]}"


def test_cached_arena_check_notices_what_the_full_walk_noticed():
    """ParamArena.ensure() runs in front of every forward pass; since round 6 its common answer ("nothing moved") comes from a cached list
    (one identity test + one data_ptr() per parameter) instead of a walk over the module tree.  Everything the walk caught must still
    trigger a rebuild: a parameter's storage replaced (`p.data = ..`), the module moved / re-typed (`.double()` and back), a Parameter
    object replaced; and `attach_grads` re-attaches gradient views after `zero_grad(set_to_none=True)`."""
    import torch
    from tpgsr_amd.engine import ParamArena
    from tpgsr_amd.model.nn_params import BatchNormParams, Conv2dParams

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = Conv2dParams(4, 8, 3, padding=1)
            self.b = BatchNormParams(8)
            self.c = torch.nn.Sequential(Conv2dParams(8, 8, 1), BatchNormParams(8))

    cpu = torch.device("cpu")
    net = Net()
    want = {n: p.detach().clone() for n, p in net.named_parameters()}
    ar = ParamArena(net)
    assert ar.ensure(cpu) is True
    assert ar.ensure(cpu) is False and ar.ensure(cpu) is False            # cached answer
    base = ar.flat.data_ptr()
    assert all(p.data_ptr() == base + 4 * ar.offsets[n] for n, p in net.named_parameters())
    # (1) one parameter's storage replaced behind the arena's back
    net.c[0].weight.data = net.c[0].weight.data.clone() * 2
    want["c.0.weight"] *= 2
    assert ar.ensure(cpu) is True and ar.ensure(cpu) is False
    # (2) the module re-typed and back (every parameter gets new storage)
    net.double().float()
    assert ar.ensure(cpu) is True and ar.ensure(cpu) is False
    # (3) a Parameter object replaced
    net.a.bias = torch.nn.Parameter(torch.full((8,), 0.5))
    want["a.bias"] = torch.full((8,), 0.5)
    assert ar.ensure(cpu) is True and ar.ensure(cpu) is False
    base = ar.flat.data_ptr()
    for n, p in net.named_parameters():
        assert p.data_ptr() == base + 4 * ar.offsets[n] and torch.equal(p.detach(), want[n]), n
        assert p.grad is not None and p.grad.data_ptr() == ar.grad.data_ptr() + 4 * ar.offsets[n]
    # gradient views: dropped by zero_grad(set_to_none=True), re-attached (and the arena zeroed) by attach_grads
    ar.grad.fill_(3.0)
    assert ar.attach_grads() is False and float(ar.grad.sum()) == 3.0 * ar.grad.numel()
    torch.optim.SGD(net.parameters(), lr=0.1).zero_grad(set_to_none=True)
    assert all(p.grad is None for p in net.parameters())
    assert ar.attach_grads() is True and float(ar.grad.abs().sum()) == 0.0
    assert all(p.grad.data_ptr() == ar.grad.data_ptr() + 4 * ar.offsets[n] for n, p in net.named_parameters())


_COPY = r'''
import copy, io, json, sys, torch
sys.path.insert(0, %(root)r)
from oracle import tpgsr_oracle as O
from tpgsr_amd import kernels as K
assert K.DRYRUN
from tpgsr_amd.model import tsrn
from tpgsr_amd.model.crnn import crnn
out = {}
lr, hr = O.synthetic_batch(2, 1)
for tag, net, x in (("tsrn", tsrn.TSRN(STN=True, mask=True).train(), lr), ("crnn", crnn.CRNN(32, 1, 37, 256).train(), torch.rand(2, 1, 32, 100))):
    net(x)                                             # the engine exists now: plans, ctypes argument blocks, the arena
    assert "_eng" in net.__dict__
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    twin = copy.deepcopy(net)
    buf = io.BytesIO()
    torch.save(net, buf)
    buf.seek(0)
    loaded = torch.load(buf, weights_only=False)
    res = {}
    for name, m in (("deepcopy", twin), ("torch.save/load", loaded)):
        same = all(torch.equal(v, sd[k]) for k, v in m.state_dict().items()) and list(m.state_dict()) == list(sd)
        own = all(a.data_ptr() != b.data_ptr() for a, b in zip(m.parameters(), net.parameters()))
        m(x)                                           # builds its own engine
        res[name] = dict(no_engine_carried=True, same_state=same, own_storage=own, own_engine=m._engine() is not net._engine())
    res["original_untouched"] = net._engine().arena.ensure(x.device) is False
    res["file_bytes"] = buf.getbuffer().nbytes
    res["param_bytes"] = 4 * sum(p.numel() for p in net.parameters())
    out[tag] = res
print("JSON" + json.dumps(out))
'''


@pytest.mark.timeout(900)
def test_modules_with_an_engine_can_be_copied_and_saved_whole():
    """copy.deepcopy(model), pickle and torch.save(model) of a network that has already run: the HIP engine (ctypes argument blocks, plans,
    workspaces) is process-local and stays behind -- the copy carries parameters and buffers, owns its storage and builds its own engine on
    first use; a whole-module file is about the size of the parameters (the arena is stored once, not once per view).  Before round 6:
    "ValueError: ctypes objects containing pointers cannot be pickled"."""
    res = _run(_COPY)
    for tag in ("tsrn", "crnn"):
        r = res[tag]
        for how in ("deepcopy", "torch.save/load"):
            assert r[how] == dict(no_engine_carried=True, same_state=True, own_storage=True, own_engine=True), (tag, how, r)
        assert r["original_untouched"], r
        assert r["file_bytes"] < 1.3 * r["param_bytes"] + (1 << 20), r
