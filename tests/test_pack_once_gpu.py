"""GPU: eval-mode networks pack / split their MFMA operands only when the parameters changed (engine._EngineBase.pack_if_stale) -- the
frozen teacher recogniser of the TPGSR step (interfaces/super_resolution.py:165, `.eval()`), everything under TextSREvaluator.
Every way the parameters can change must be noticed: writes through tensors (load_state_dict, copy_) by the arena's version counter,
the fused Adam kernel (raw pointer) by the engine's own counter."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tpgsr_oracle as O  # noqa: E402

DEV = "cuda"


def _crnn(seed):
    from tpgsr_amd.model.crnn import crnn
    sd = O.recipe_state_dict(O.crnn_spec(), seed)
    m = crnn.CRNN(32, 1, 37, 256)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


def _count_packs(net):
    eng = net._engine()
    runs = {"n": 0}
    for pl in eng._plans.values():
        p = pl["pack"]
        if getattr(p, "_counted", False):
            continue
        orig = p.run

        def run(orig=orig):
            runs["n"] += 1
            orig()
        p.run, p._counted = run, True
    return runs


def test_eval_network_repacks_exactly_when_its_parameters_change():
    gray = torch.rand(4, 1, 32, 100, generator=torch.Generator().manual_seed(1)).to(DEV)
    a, sd_a = _crnn(5)
    with torch.no_grad():
        y0 = a(gray).clone()
        runs = _count_packs(a)
        y1 = a(gray).clone()
        y2 = a(gray).clone()
    assert runs["n"] == 0 and torch.equal(y0, y1) and torch.equal(y0, y2)      # unchanged parameters: no pack, same bits
    b, sd_b = _crnn(6)
    with torch.no_grad():
        yb = b(gray).clone()
        a.load_state_dict(sd_b)                                                # through tensors: the version counter moves
        y3 = a(gray).clone()
        assert runs["n"] == 1 and torch.equal(y3, yb)
        next(a.parameters()).mul_(1.0)                                         # any in-place write counts, even a no-op
        a(gray)
        assert runs["n"] == 2
        a(gray)
        assert runs["n"] == 2


def test_train_then_eval_sees_the_updated_parameters():
    """Adam writes the arena through a raw pointer (no version bump): the engine's own counter makes the next eval forward re-pack"""
    from tpgsr_amd.interfaces.super_resolution import TSRNTrainStep
    from tpgsr_amd.model import tsrn
    sd = O.recipe_state_dict(O.tsrn_spec(STN=False, mask=True, srb_nums=2), 77)
    net = tsrn.TSRN(STN=False, mask=True, srb_nums=2)
    net.load_state_dict(sd)
    net = net.to(DEV)
    lr, hr = O.synthetic_batch(4, 3)
    lr, hr = lr.to(DEV), hr.to(DEV)
    net.eval()
    with torch.no_grad():
        e0 = net(lr).clone()
    net.train()
    ts = TSRNTrainStep(net)
    ts.step(lr, hr)
    ts.step(lr, hr)
    net.eval()
    with torch.no_grad():
        e1 = net(lr).clone()
    fresh = tsrn.TSRN(STN=False, mask=True, srb_nums=2)
    fresh.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    fresh = fresh.to(DEV).eval()
    with torch.no_grad():
        e2 = fresh(lr).clone()
    assert not torch.equal(e0, e1)
    assert torch.equal(e1, e2)


def test_graph_replay_then_eval_sees_the_updated_parameters():
    """a hipGraph replay runs the captured Adam and bumps no counter by itself -- capture, eval (packs, stores the key),
    replay x N, eval again must see the parameters the LAST replay left, not the operands packed one optimiser step earlier"""
    from tpgsr_amd.interfaces.super_resolution import TSRNTrainStep
    from tpgsr_amd.model import tsrn
    sd = O.recipe_state_dict(O.tsrn_spec(STN=False, mask=True, srb_nums=2), 78)
    net = tsrn.TSRN(STN=False, mask=True, srb_nums=2)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    lr, hr = O.synthetic_batch(4, 9)
    lr, hr = lr.to(DEV), hr.to(DEV)
    ts = TSRNTrainStep(net)
    ts.capture(lr, hr, warmup=1)
    net.eval()
    with torch.no_grad():
        e0 = net(lr).clone()                  # packs the operands of the post-warm-up parameters
    net.train()
    for _ in range(3):
        ts.replay()
    net.eval()
    with torch.no_grad():
        e1 = net(lr).clone()
    # a fresh network holding the same state (packs from scratch) is the truth
    fresh = tsrn.TSRN(STN=False, mask=True, srb_nums=2)
    fresh.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    fresh = fresh.to(DEV).eval()
    with torch.no_grad():
        e2 = fresh(lr).clone()
    assert not torch.equal(e0, e1)
    assert torch.equal(e1, e2)
    # invalidate_packed(): a write through `p.data` bumps no version counter -- the documented call makes the next forward re-pack
    with torch.no_grad():
        next(net.parameters()).data.mul_(1.5)
        stale = net(lr).clone()
        net._engine().invalidate_packed()
        fresh_out = net(lr).clone()
    assert torch.equal(stale, e1) and not torch.equal(fresh_out, e1)


def test_pack_check_catches_a_write_behind_the_version_counters(monkeypatch):
    """TPGSR_PACK_CHECK=1 (debugging aid): a skipped pack whose arena no longer holds the packed bits raises instead of
    running on stale operands; after invalidate_packed() the forward packs again and passes"""
    monkeypatch.setenv("TPGSR_PACK_CHECK", "1")
    gray = torch.rand(2, 1, 32, 100, generator=torch.Generator().manual_seed(3)).to(DEV)
    a, _ = _crnn(9)
    with torch.no_grad():
        y0 = a(gray).clone()
        assert torch.equal(a(gray), y0)                      # unchanged: the check passes silently
        next(a.parameters()).data.mul_(1.25)                 # no version counter moves
        with pytest.raises(RuntimeError, match="invalidate_packed"):
            a(gray)
        a._engine().invalidate_packed()
        y1 = a(gray).clone()
    assert not torch.equal(y0, y1)
