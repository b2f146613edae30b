"""GPU, ONE rank over RCCL (backend "nccl" on ROCm): the data-parallel TPGSR step with its collectives FORCED at world size 1
(`force_collectives=True`) -- process-group init with device_id, the asynchronous bucket all-reduce launched inside the backward pass
(RCCL's own stream against the step's three streams), `wait()`, the averaging kernel, clip + Adam -- must be bitwise the plain world-1
step (one rank: the all-reduce is the identity, the average a multiplication by 1.0).  This is the part of SURVEY.md section 8(e) a
one-GPU box CAN execute: the code path of the 8-GPU job (reference: nn.DataParallel, interfaces/base.py:394-400) with RCCL really
running.  Own process: a failed RCCL bring-up must not take the rest of the GPU suite with it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

_SCRIPT = r'''
import os, sys
sys.path.insert(0, %(root)r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%(port)d), HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from oracle import tpgsr_oracle as O
from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep, TSRNTrainStep
from tpgsr_amd.model import tsrn
from tpgsr_amd.model.crnn import crnn

def build():
    sr = tsrn.TSRN_TL(STN=True, mask=True)
    sr.load_state_dict(O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 11, tps_hw=(16, 64)))
    teacher = crnn.CRNN(32, 1, 37, 256)
    teacher.load_state_dict(O.recipe_state_dict(O.crnn_spec(), 12))
    stu = crnn.CRNN(32, 1, 37, 256)
    stu.load_state_dict(O.recipe_state_dict(O.crnn_spec(), 13))
    return sr.to(dev).train(), stu.to(dev).train(), teacher.to(dev).eval()

lr, hr = O.synthetic_batch(6, 1234)
lr, hr = lr.to(dev), hr.to(dev)
res = []
for force in (False, True):
    sr, stu, teacher = build()
    ts = TPGSRTrainStep([sr], [stu], teacher, stu_iter=1, world_size=1, force_collectives=force)
    ts.broadcast_parameters(0)
    losses = [float(ts.step(lr, hr).item()) for _ in range(3)]
    torch.cuda.synchronize()
    if force:
        ex = ts._exchanger()
        # SR networks | the text-prior generator from conv3 on (final between its two backward plans) | its first three conv layers
        assert ex.active and ex.world == 1 and len(ex.bounds) == 3
        assert ex.bounds[2] == (ex.bounds[0][1], ex.bounds[1][0]) and not ex._work
    res.append((losses, ts.pool.flat.clone(), ts.pool.grad.clone()))
(l0, p0, g0), (l1, p1, g1) = res
assert l0 == l1, (l0, l1)
assert torch.equal(g0, g1), float((g0 - g1).abs().max())
assert torch.equal(p0, p1), float((p0 - p1).abs().max())
# three text-prior generators (the C5 shape): every student its own early bucket -- 1 + 2 * 3 buckets, launched from the
# weight-gradient stream at the end of each plan that finishes a range (order pinned on the recorded plans by
# tests/test_plan_dryrun_cpu.py) -- bitwise the plain step again
res = []
for force in (False, True):
    sr = tsrn.TSRN_TL(STN=True, mask=True)
    sr.load_state_dict(O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 21, tps_hw=(16, 64)))
    srs, stus = [sr.to(dev).train()], []
    for i in range(3):
        stu = crnn.CRNN(32, 1, 37, 256)
        stu.load_state_dict(O.recipe_state_dict(O.crnn_spec(), 31 + i))
        stus.append(stu.to(dev).train())
    teacher = crnn.CRNN(32, 1, 37, 256)
    teacher.load_state_dict(O.recipe_state_dict(O.crnn_spec(), 12))
    ts = TPGSRTrainStep(srs, stus, teacher.to(dev).eval(), stu_iter=3, sr_share=True, world_size=1, force_collectives=force)
    ts.broadcast_parameters(0)
    losses = [float(ts.step(lr, hr).item()) for _ in range(3)]
    torch.cuda.synchronize()
    if force:
        ex = ts._exchanger()
        assert ex.active and len(ex.bounds) == 7 and not ex._work
        cover = sorted(ex.bounds)
        assert cover[0][0] == 0 and all(a[1] == b[0] for a, b in zip(cover, cover[1:])), cover
    res.append((losses, ts.pool.flat.clone(), ts.pool.grad.clone()))
(l0, p0, g0), (l1, p1, g1) = res
assert l0 == l1, (l0, l1)
assert torch.equal(g0, g1), float((g0 - g1).abs().max())
assert torch.equal(p0, p1), float((p0 - p1).abs().max())
# the C2 driver (one flat bucket after the backward pass)
res = []
for force in (False, True):
    net = tsrn.TSRN(STN=True, mask=True)
    net.load_state_dict(O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True), 1234, tps_hw=(16, 64)))
    ts = TSRNTrainStep(net.to(dev).train(), world_size=1, force_collectives=force)
    for _ in range(2):
        ts.step(lr, hr)
    torch.cuda.synchronize()
    res.append(ts.pool.flat.clone())
assert torch.equal(res[0], res[1])
# one raw all-reduce through the exchanger on a buffer with a known content: RCCL really touched it
from tpgsr_amd.distributed import GradientExchanger
flat = torch.arange(1 << 20, dtype=torch.float32, device=dev)
ex = GradientExchanger(flat, [(0, 1 << 19), (1 << 19, 1 << 20)], force=True)
ex.launch(0)
ex.finish()
torch.cuda.synchronize()
assert torch.equal(flat, torch.arange(1 << 20, dtype=torch.float32, device=dev))
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
'''


@pytest.mark.timeout(420)
def test_forced_collectives_over_rccl_equal_the_plain_step():
    port = 33500 + os.getpid() % 2000
    out = subprocess.run([sys.executable, "-c", _SCRIPT % dict(root=ROOT, port=port)], capture_output=True, text=True, timeout=400,
                         cwd=ROOT)
    assert out.returncode == 0 and "RCCL_WORLD1_OK" in out.stdout, out.stdout[-2000:] + "\n" + out.stderr[-4000:]
