"""GPU: the MORAN evaluation recognizer on the HIP kernels (tpgsr_amd/model/moran) against the fixture generated from the imported
reference (tests/golden/make_golden_moran.py) and against oracle/moran_oracle.py.  Tolerances: the x3 arithmetic is fp32-equivalent
(~1e-6 relative per GEMM).  Measured on MI355X: rectified image 3.3e-6 abs (a NOISE image re-sampled at positions that carry the
offset CNN's rounding), ResNet features 1.4e-6 relative, recurrent features 4.5e-6, decoder scores 4e-7 relative from the reference's
own features and 2.3e-6 end to end; the gates below leave a factor of ~5-10.  The arg-max decisions (top-2 margin > 2.5e-3 in the
fixture) must be identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import moran_oracle as M  # noqa: E402
from tests.moran_common import fixture, state_dict  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def net():
    from tpgsr_amd.model.moran import MORAN
    g = fixture()
    m = MORAN(1, 37, 256, 32, 100, BidirDecoder=True).eval()
    m.load_state_dict(state_dict(int(g["seed"])), strict=True)
    return m.to(DEV), g


def _rel(a, b):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def test_parse_moran_data(net):
    from tpgsr_amd.interfaces.super_resolution import parse_moran_data
    _, g = net
    x, length, text, text_rev = parse_moran_data(torch.tensor(g["sr"]).to(DEV))
    assert (x.cpu() - torch.tensor(g["x"])).abs().max().item() < 2e-6
    assert length.tolist() == [20, 20, 20] and text.numel() == 60 and int(text.abs().sum()) == 0 and text_rev is text


def test_rectifier_vs_reference(net):
    m, g = net
    rect = m.MORN(torch.tensor(g["x"]).to(DEV), True)
    err = (rect.cpu() - torch.tensor(g["rectified"])).abs().max().item()
    print(f"MORN rectified image: max abs err {err:.2e}")
    assert err < 2e-5
    with pytest.raises(NotImplementedError):
        m.MORN(torch.zeros(1, 1, 32, 128, device=DEV), True)


def test_recognizer_stages_from_reference_intermediates(net):
    """ASRN from the reference's own rectified image; the decoders from the reference's own recurrent features"""
    m, g = net
    conv, rnn = m.ASRN.features(torch.tensor(g["rectified"]).to(DEV))
    e_conv = _rel(conv.permute(0, 3, 1, 2), g["conv"])
    e_rnn = (rnn.cpu() - torch.tensor(g["rnn"]).permute(1, 0, 2)).abs().max().item()
    print(f"ASRN conv features rel err {e_conv:.2e}, recurrent features abs err {e_rnn:.2e}")
    assert e_conv < 2e-5 and e_rnn < 2e-5
    length = torch.tensor([20, 20, 20])
    feats = torch.tensor(g["rnn"]).permute(1, 0, 2).contiguous().to(DEV)
    for name, key in (("attentionL2R", "l2r"), ("attentionR2L", "r2l")):
        out = getattr(m.ASRN, name)(feats, length, None, test=True)
        e = _rel(out, g[key])
        print(f"{name}: logits rel err {e:.2e}")
        assert e < 5e-6
        assert torch.equal(out.argmax(1).cpu(), torch.tensor(g[key]).argmax(1))


def test_end_to_end_vs_reference_and_oracle(net):
    from tpgsr_amd.interfaces.super_resolution import TextSREvaluator, parse_moran_data
    from tpgsr_amd.utils.metrics import get_string_moran
    m, g = net
    sr = torch.tensor(g["sr"]).to(DEV)
    x, length, text, text_rev = parse_moran_data(sr)
    l2r, r2l = m(x, length, text, text_rev, test=True)
    (l2r_d, r2l_d), none = m(x, length, text, text_rev, test=True, debug=True)
    assert none is None and torch.equal(l2r_d, l2r) and torch.equal(r2l_d, r2l)          # deterministic, same predictions with debug
    e1, e2 = _rel(l2r, g["l2r"]), _rel(r2l, g["r2l"])
    print(f"end to end: l2r rel err {e1:.2e}, r2l {e2:.2e}; min top-2 margin of the fixture {float(g['margin'].min()):.2e}")
    assert e1 < 3e-5 and e2 < 3e-5
    assert torch.equal(l2r.argmax(1).cpu(), torch.tensor(g["l2r"]).argmax(1))
    assert torch.equal(r2l.argmax(1).cpu(), torch.tensor(g["r2l"]).argmax(1))
    strings = get_string_moran(l2r, length)
    assert strings == list(g["strings"])
    # the oracle on the same weights, different input (a second batch the fixture does not hold)
    p = state_dict(int(g["seed"]))
    g2 = torch.Generator().manual_seed(5)
    sr2 = torch.rand(4, 3, 16, 64, generator=g2)
    xo, lo, _, _ = M.parse_moran_data(sr2)
    with torch.no_grad():
        o = M.moran(p, xo, lo)
    ev = TextSREvaluator([], [], recognizer=m)
    x2, l2, t2, _ = parse_moran_data(sr2.to(DEV))
    got, _ = m(x2, l2, t2, t2, test=True)
    mg = torch.softmax(o["l2r"], 1).topk(2, 1).values
    clear = (mg[:, 0] - mg[:, 1]) > 1e-3                         # decisions the oracle itself takes with a clear margin
    assert torch.equal(got.argmax(1).cpu()[clear], o["l2r"].argmax(1)[clear]) and int(clear.sum()) > 40
    assert _rel(got, o["l2r"]) < 1e-3
    if bool(clear.all()):
        assert ev.recognize(sr2.to(DEV)) == M.get_string_moran(o["l2r"], lo)
    else:
        assert len(ev.recognize(sr2.to(DEV))) == 4
