"""CPU: the ASTER evaluation oracle (oracle/aster_oracle.py) against the fixture generated from the imported reference
(tests/golden/make_golden_aster.py), and the host-side string decode."""
import torch

from oracle import aster_oracle as A
from tests.aster_common import fixture, state_dict


def test_oracle_matches_reference_fixture():
    g = fixture()
    p = state_dict(int(g["seed"]))
    voc = A.get_vocabulary("all")
    lr = torch.tensor(g["lr"])
    images = A.parse_aster_data(lr)
    assert torch.equal(images, torch.tensor(g["images"]))
    with torch.no_grad():
        o = A.aster_greedy(p, images, len(voc), int(g["max_len"]))
    for k, tol in (("ctrl", 1e-6), ("rectified", 1e-6), ("feats", 5e-6), ("scores", 5e-6)):
        assert (o[k] - torch.tensor(g[k])).abs().max().item() <= tol, k
    assert torch.equal(o["ids"], torch.tensor(g["ids"]))


def test_vocabulary_and_string_decode():
    voc = A.get_vocabulary("all")
    assert len(voc) == 10 + 52 + 32 + 3 and voc[-3:] == ["EOS", "PADDING", "UNKNOWN"]
    assert len(A.get_vocabulary("lower")) == 39
    eos, unk = voc.index("EOS"), voc.index("UNKNOWN")
    ids = torch.tensor([[voc.index("h"), voc.index("i"), unk, voc.index("!"), eos, voc.index("x")], [eos, 1, 2, 3, 4, 5]])
    assert A.get_string_aster(ids, voc) == ["hi!", ""]
