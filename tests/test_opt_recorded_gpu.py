"""GPU: the `--tpg OPT` recogniser (reference model/crnn/model.py:25-95) as a RECORDED plan (tpgsr_amd/engine_functional.py: the module's
forward and its autograd backward traced once into kernels.Plan, replayed by the native executor) against the same network run operator
by operator through autograd: logits, d gray and every parameter gradient on FRESH inputs of later replays -- a kernel that ran only at
trace time (an ATen op autograd slipped in, a weight packed once) shows up as stale data here -- plus BatchNorm's running statistics and
batch counters after several passes, and the eval-mode (teacher) plan."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_opt_student_cpu import OPT, opt_state_dicts  # noqa: E402

DEV = "cuda"


def _model(train=True):
    from tpgsr_amd.model.crnn import model as opt
    m = opt.Model(OPT)
    m.load_state_dict(opt_state_dicts()[1])
    m = m.to(DEV)
    return m.train() if train else m.eval()


def _passes(record, N=6, n_pass=3, train=True):
    m = _model(train)
    eng = m._engine()
    eng.record = record
    g = torch.Generator().manual_seed(77)
    outs = []
    for i in range(n_pass):
        gray = torch.rand(N, 1, 32, 100, generator=g).to(DEV)
        dlog = torch.randn(N, 26, 37, generator=g).to(DEV)
        logits = eng.forward(gray, train)
        o = dict(logits=logits.clone())
        if train:
            eng.arena.attach_grads()
            eng.arena.grad.zero_()
            o["dgray"] = eng.backward(N, gray, dlog, need_dgray=True).clone()
            o["grad"] = eng.arena.grad.clone()
        torch.cuda.synchronize()
        outs.append(o)
    eng.flush_counters()
    bufs = {k: v.clone() for k, v in m.named_buffers()}
    return outs, bufs, eng


def test_recorded_plan_equals_operator_by_operator_run():
    rec, bufs_r, eng = _passes(True)
    ref, bufs_e, _ = _passes(False)
    assert eng._plans and all(len(pl["fwd"]) > 100 and len(pl["bwd"]) > 200 for pl in eng._plans.values())
    for i, (a, b) in enumerate(zip(rec, ref)):
        assert torch.equal(a["logits"], b["logits"]), (i, float((a["logits"] - b["logits"]).abs().max()))
        assert torch.equal(a["dgray"], b["dgray"]), (i, float((a["dgray"] - b["dgray"]).abs().max()))
        d = (a["grad"] - b["grad"]).abs().max().item()
        assert d <= 1e-6 * b["grad"].abs().max().item(), (i, d)
    for k in bufs_e:
        assert torch.equal(bufs_r[k].float(), bufs_e[k].float()), k


def test_recorded_eval_plan_follows_parameter_updates():
    """the frozen teacher / evaluation: an eval-mode plan re-packs its weights at every replay, so it sees a checkpoint load"""
    m = _model(False)
    eng = m._engine()
    g = torch.Generator().manual_seed(5)
    gray = torch.rand(4, 1, 32, 100, generator=g).to(DEV)
    a = eng.forward(gray, False).clone()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.01)
    b = eng.forward(gray, False).clone()
    eng.record = False
    c = eng.forward(gray, False)
    torch.cuda.synchronize()
    assert not torch.equal(a, b)
    assert torch.equal(b, c)
