"""CPU: the plan-recording host logic of every engine / train-step driver, without a GPU (TPGSR_PLAN_DRYRUN=1, see
tpgsr_amd/kernels.py): each wrapper's argument list is checked against the C-ABI signature, the recorded plans are handed
to the native executor (entry point + argument count), and the recorded launch geometry is inspected -- nothing is
computed.  Runs in a subprocess so the dry-run switch never leaks into other tests."""
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
from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep, TSRNTrainStep
from tpgsr_amd.model import tsrn
from tpgsr_amd.model.crnn import crnn

def conv_stats(nets):
    n, scalar, names = 0, [], {}
    for net in nets:
        for pl in net._engine()._plans.values():
            for plan in [v for k, v in pl.items() if k != "ws"]:
                if len(plan):
                    plan.run()                      # builds the native plan: validates symbol + argument count
                for name, fn, args, sid in plan.ops:
                    names[name] = names.get(name, 0) + 1
                    if name in ("tpgsr_conv_fwd", "tpgsr_conv_wgrad"):
                        a = args[0]._obj
                        c = a.c if name == "tpgsr_conv_wgrad" else a
                        n += 1
                        if name == "tpgsr_conv_fwd" and c.bnb_y:
                            names["conv_fwd+bnb"] = names.get("conv_fwd+bnb", 0) + 1
                        wld = c.wt_ld if c.wt_ld > 0 else c.Cout
                        if c.Cin %% 4 != 0 or (name == "tpgsr_conv_fwd" and wld %% 4 != 0) or \
                           (name == "tpgsr_conv_wgrad" and not a.dy_ps and a.dy_ld %% 4 != 0):
                            scalar.append((name, c.Cin, c.Cout, c.KH, c.KW))
    return n, scalar, names

out = {}
N = 4
lr, hr = O.synthetic_batch(N, 1)
# C2
net = tsrn.TSRN(STN=True, mask=True).train()
ts = TSRNTrainStep(net)
ts.step(lr, hr)
out["c2"] = conv_stats([net])
# C3 / C5 shape
sr = tsrn.TSRN_TL(STN=True, mask=True).train()
teacher = crnn.CRNN(32, 1, 37, 256).eval()
stus = [crnn.CRNN(32, 1, 37, 256).train() for _ in range(3)]
ts = TPGSRTrainStep([sr], stus[:1], teacher, stu_iter=1)
ts.step(lr, hr)
out["c3"] = conv_stats([sr, stus[0], teacher])
spl = [pl for pl in stus[0]._engine()._plans.values() if "bwd" in pl and len(pl["bwd"])][0]
out["stu_bwd_split"] = [[op[0] for op in spl[k].ops if op[1] is None] for k in ("bwd", "bwd_b")] + [stus[0]._engine().early_final_offset()]
sr5 = tsrn.TSRN_TL(STN=True, mask=True).train()
ts5 = TPGSRTrainStep([sr5], stus, teacher, stu_iter=3, sr_share=True)
ts5.step(lr, hr)
out["c5"] = conv_stats([sr5] + stus + [teacher])
out["pool"] = [ts5.pool.flat.numel(), sum(ts5.pool.ranges[id(m)][1] - ts5.pool.ranges[id(m)][0] for m in ts5.pool.modules)]
# parameters are views of ONE buffer, SR net first
base = ts5.pool.flat.data_ptr()
out["views"] = all(base <= p.data_ptr() < base + 4 * ts5.pool.flat.numel() for m in [sr5] + stus for p in m.parameters())
out["sr_first"] = ts5.pool.ranges[id(sr5)][0] == 0
# module API (autograd) in dry-run: slots are acquired and released
x = lr.clone()
y = sr5(x, torch.zeros(N, 37, 1, 26))
y.sum().backward()
out["live_after_bwd"] = len(sr5._engine()._live)
# a training-mode forward whose graph is dropped without a backward pass gives its workspace slot back
import gc
for _ in range(3):
    y = sr5(x, torch.zeros(N, 37, 1, 26))
    del y
gc.collect()
out["live_after_dropped_graphs"] = len(sr5._engine()._live)
# a module handed to a SECOND pool is re-pointed there; the first pool notices on its next bind instead of using a stale buffer
from tpgsr_amd.engine import ArenaPool
p1 = ts5.pool
p2 = ArenaPool([sr5])
p2.bind(torch.device("cpu"))
out["repointed"] = sr5._engine().arena.flat.data_ptr() == p2.flat.data_ptr()
p1.bind(torch.device("cpu"))
out["first_pool_rebuilt"] = sr5._engine().arena.flat.data_ptr() == p1.flat.data_ptr() + 4 * p1.ranges[id(sr5)][0]
# C5 with the gradient exchange on: every text-prior generator has its own early bucket (conv3 .. end, launched between the two plans
# of ITS backward pass) and a rest bucket (after its second plan); the shared SR network's bucket leaves after the last SR backward
# (stage 0).  The exchanger is inactive here (no process group): its launch calls are recorded.
sr6 = tsrn.TSRN_TL(STN=True, mask=True).train()
stus6 = [crnn.CRNN(32, 1, 37, 256).train() for _ in range(3)]
ts6 = TPGSRTrainStep([sr6], stus6, teacher, stu_iter=3, sr_share=True, world_size=1, force_collectives=True)
ts6.pool.bind(lr.device)
ts6._buffers(lr)
ex = ts6._exchanger()
order = []
ex.launch = lambda b: order.append(b)
ts6.step(lr, hr)
out["c5_bucket_order"] = order
out["c5_bounds"] = ex.bounds
out["c5_ranges"] = [list(ts6.pool.ranges[id(m)]) for m in ts6.pool.modules]
out["c5_cut"] = stus6[0]._engine().early_final_offset()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.timeout(600)
def test_record_all_plans_without_gpu():
    env = dict(os.environ, TPGSR_PLAN_DRYRUN="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], capture_output=True, text=True, env=env, timeout=550)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for cfg in ("c2", "c3", "c5"):
        n, scalar, names = res[cfg]
        assert n > 50
        # no MFMA GEMM of the training step is left on the scalar (dword-gather) loaders: 1-channel and 37-class operands
        # are zero-padded / im2col'ed to multiples of 4 (CRNN conv0, Linear(512,37), InfoGen tconv1)
        assert scalar == [], (cfg, scalar)
    assert res["c3"][2].get("tpgsr_im2col3x3_c1", 0) == 2          # student + teacher conv0
    assert res["c3"][2].get("tpgsr_pad_channels", 0) >= 2          # prior 37 -> 40, dlogits 37 -> 40
    # BatchNorm-backward sums ride on the producing data-gradient convolutions: 11 in the SR network (5 blocks x 2 + block 7), 3 in the
    # student (conv2 / conv4 / bn6), 3 in InfoGen; what is left as its own launch: InfoGen's last BatchNorm and the STN head
    # ... + the tail's mish backward (an epilogue without sums) + InfoGen bn0..bn2 + bn6 (BiLSTM projection's data gradient)
    assert res["c3"][2].get("conv_fwd+bnb", 0) == 11 + 1 + 3 + 3, res["c3"][2]
    assert res["c2"][2].get("conv_fwd+bnb", 0) == 11 + 1
    nm = res["c3"][2]
    n_apply = nm.get("tpgsr_bn_bwd_apply_bnd", 0) + nm.get("tpgsr_bn_bwd_apply", 0)
    assert nm.get("tpgsr_bn_bwd_reduce", 0) == n_apply - 17
    # (round 5's in-launch finalize, csrc/bn_derive.h, is off by default -- measured slower inside the step: every BatchNorm keeps its
    #  finalize launch; tests/test_bn_derive_gpu.py runs the switched-on plans)
    assert nm.get("tpgsr_bn_bwd_finalize", 0) == n_apply and nm.get("tpgsr_affine_act_bnd", 0) == 0
    assert res["c3"][2].get("tpgsr_act_bwd", 0) == 0
    # the text-prior generator's backward pass is two plans: the first forks and never joins, the second ends with THE join; everything
    # from conv3 on (offset 370176 of 8331304 floats) is final in between
    a_edges, b_edges, cut = res["stu_bwd_split"]
    assert "join" not in a_edges and a_edges.count("fork") >= 2 and b_edges[-1] == "join" and b_edges.count("join") == 1
    assert cut == 370176
    assert res["pool"][0] >= res["pool"][1] and res["views"] and res["sr_first"]
    assert res["live_after_bwd"] == 0
    assert res["live_after_dropped_graphs"] == 0
    assert res["repointed"] and res["first_pool_rebuilt"]
    # C5's buckets: [SR | stu0 early, stu0 rest | stu1 early, stu1 rest | stu2 early, stu2 rest]; the cascade's backward pass runs stage 2,
    # 1, 0, so stage 2's and stage 1's generators leave first, the shared SR network after its last backward pass (stage 0), then stage
    # 0's early bucket; stage 0's rest (bucket 2) is what finish() still has to launch
    assert res["c5_bucket_order"] == [5, 6, 3, 4, 0, 1], res["c5_bucket_order"]
    b, rng, cut = [tuple(x) for x in res["c5_bounds"]], res["c5_ranges"], res["c5_cut"]
    assert len(b) == 7 and b[0] == (0, rng[0][1])
    for k in range(3):
        a_k, e_k = rng[1 + k]
        assert b[1 + 2 * k] == (a_k + cut, e_k)                      # early: conv3 .. end of this generator's arena
        assert b[2 + 2 * k] == (rng[k][1], a_k + cut)                # rest: from the end of the previous slice (alignment gap included)
    cover = sorted(b)
    assert cover[0][0] == 0 and all(cover[i][1] == cover[i + 1][0] for i in range(6)) and cover[-1][1] == rng[-1][1]


FSCRIPT = r'''
import sys, torch
sys.path.insert(0, %(root)r)
from tpgsr_amd import kernels as K
assert K.DRYRUN
from tpgsr_amd.model import tsrn
from tpgsr_amd.model.stn_head import STNHead
from tpgsr_amd.model.tps_spatial_transformer import TPSSpatialTransformer
x = torch.randn(2, 64, 8, 24, requires_grad=True)
t = torch.rand(2, 32, 8, 24, requires_grad=True)
for m, inp in ((tsrn.GruBlock(64, 64), (x,)), (tsrn.RecurrentResidualBlock(64), (x,)), (tsrn.RecurrentResidualBlockTL(64, 32), (x, t)),
               (tsrn.UpsampleBLock(64, 2), (x,)), (tsrn.InfoGen(37, 32), (torch.rand(2, 37, 1, 26, requires_grad=True),)), (tsrn.mish(), (x,))):
    y = m.train()(*inp)
    y.sum().backward()
    print(type(m).__name__, tuple(y.shape))
lr = torch.rand(3, 4, 16, 64, requires_grad=True)
head, tps = STNHead(4, 20, "none", input_size=[16, 64]).train(), TPSSpatialTransformer((16, 64), 20, (0.05, 0.05))
feat, ctrl = head(lr)
y, src = tps(lr, ctrl)
y.sum().backward()
print("STN", tuple(feat.shape), tuple(ctrl.shape), tuple(y.shape), tuple(src.shape))
print("RESULT ok")
'''


@pytest.mark.timeout(600)
def test_standalone_blocks_record_without_gpu():
    """every standalone block forward + backward issues well-formed C-ABI calls (argument lists checked, nothing computed)"""
    env = dict(os.environ, TPGSR_PLAN_DRYRUN="1")
    r = subprocess.run([sys.executable, "-c", FSCRIPT % dict(root=ROOT)], capture_output=True, text=True, env=env, timeout=550)
    assert r.returncode == 0 and "RESULT ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "UpsampleBLock (2, 64, 16, 48)" in r.stdout and "InfoGen (2, 32, 1, 203)" in r.stdout
    assert "STN (3, 512) (3, 20, 2) (3, 4, 16, 64) (3, 1024, 2)" in r.stdout


NSCRIPT = r'''
import sys, torch
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
from tpgsr_amd import kernels as K
assert K.DRYRUN
import test_next_models_gpu as T
lr = torch.rand(2, 4, 16, 64, requires_grad=True)
prior = torch.rand(2, 37, 1, 26, requires_grad=True)
for name in ("srresnet_tl", "srcnn_tl", "vdsr_tl", "rdn_tl"):
    net = T._build(name).train()
    y = net(lr, prior)
    y.sum().backward()
    assert tuple(y.shape) == (2, 4, 32, 128), (name, y.shape)
    net.eval()
    with torch.no_grad():
        assert tuple(net(lr, prior).shape) == (2, 4, 32, 128)
net = T._build("opt").train()
y = net(torch.rand(2, 1, 32, 100, requires_grad=True))
y.sum().backward()
assert tuple(y.shape) == (26, 2, 37), y.shape
print("RESULT ok")
'''


@pytest.mark.timeout(600)
def test_next_models_record_without_gpu():
    """the N3 / N4 networks issue well-formed C-ABI calls end to end (forward + backward), shapes as the reference's"""
    env = dict(os.environ, TPGSR_PLAN_DRYRUN="1")
    r = subprocess.run([sys.executable, "-c", NSCRIPT % dict(root=ROOT)], capture_output=True, text=True, env=env, timeout=550)
    assert r.returncode == 0 and "RESULT ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
