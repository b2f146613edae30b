"""CPU: the MORAN evaluation oracle (oracle/moran_oracle.py) against the fixture generated from the imported reference
(tests/golden/make_golden_moran.py), the host-side string decode, and the drop-in module's state_dict contract."""
import pytest
import torch

from oracle import moran_oracle as M
from tests.moran_common import fixture, layout, state_dict


def test_oracle_matches_reference_fixture():
    g = fixture()
    p = state_dict(int(g["seed"]))
    x, length, text, _ = M.parse_moran_data(torch.tensor(g["sr"]))
    assert torch.equal(x, torch.tensor(g["x"])) and length.tolist() == [20, 20, 20] and text.numel() == 60
    with torch.no_grad():
        o = M.moran(p, x, length)
    for k, tol in (("offsets", 1e-6), ("rectified", 1e-6), ("conv", 2e-4), ("rnn", 5e-6), ("l2r", 5e-5), ("r2l", 5e-5)):
        assert (o[k] - torch.tensor(g[k])).abs().max().item() <= tol, k          # (conv features reach 240, logits 19)
    assert torch.equal(o["l2r"].argmax(1), torch.tensor(g["l2r"]).argmax(1))
    assert M.get_string_moran(o["l2r"], length) == list(g["strings"])


def test_string_decode():
    from tpgsr_amd.utils.metrics import MORAN_ALPHABET, get_string_moran, moran_decode
    assert MORAN_ALPHABET == M.alphabet() and len(MORAN_ALPHABET) == 37
    abc = MORAN_ALPHABET
    ids = [abc.index(c) for c in "hi$xx" + "$abcd" + "nodollar"]
    assert moran_decode(ids, [5, 5, 8]) == ["hi$xx", "$abcd", "nodollar"] == M.decode(torch.tensor(ids), torch.tensor([5, 5, 8]))
    logits = torch.full((18, 37), -1.0)
    logits[torch.arange(18), torch.tensor(ids)] = 1.0
    assert get_string_moran(logits, torch.tensor([5, 5, 8])) == ["hi", "", "nodollar"]
    with pytest.raises(ValueError):
        moran_decode(ids, [5, 5])


def test_module_state_dict_contract_and_no_cpu_path():
    from tpgsr_amd.model.moran import MORAN
    m = MORAN(1, 37, 256, 32, 100, BidirDecoder=True).eval()
    assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == [(k, list(s)) for k, s in layout()]
    m.load_state_dict(state_dict(5151), strict=True)
    x = torch.zeros(2, 1, 32, 100)
    with pytest.raises(RuntimeError, match="GPU only|no CPU"):
        m(x, torch.tensor([20, 20]), None, None, test=True)
    with pytest.raises(RuntimeError, match="evaluation"):
        m.train()(x, torch.tensor([20, 20]), None, None, test=False)


def test_host_logic_dry_run():
    """the whole HIP-path forward with TPGSR_PLAN_DRYRUN=1 (argument lists checked against the C ABI, nothing computed)"""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from tpgsr_amd import kernels as K; assert K.DRYRUN\n"
            "from tpgsr_amd.model.moran import MORAN\n"
            "m = MORAN(1, 37, 256, 32, 100, BidirDecoder=True).eval()\n"
            "l2r, r2l = m(torch.rand(3, 1, 32, 100), torch.tensor([20, 7, 12]), None, None, test=True)\n"
            "assert tuple(l2r.shape) == (39, 37) and tuple(r2l.shape) == (39, 37)\n"
            "m1 = MORAN(1, 37, 256, 32, 100, BidirDecoder=False).eval()\n"
            "assert tuple(m1(torch.rand(2, 1, 32, 100), torch.tensor([5, 5]), None, None, test=True).shape) == (10, 37)\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, TPGSR_PLAN_DRYRUN="1"), timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
