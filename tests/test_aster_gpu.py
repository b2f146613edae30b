"""GPU: the ASTER evaluation recognizer on the HIP path (tpgsr_amd/model/recognizer, csrc/aster.hip) against the fixture generated from
the imported reference and against the oracle, stage by stage: parse_aster_data, control points, rectified image, encoder features
(ResNet_ASTER + two-layer BiLSTM), greedy ids / scores; state_dict layout; evaluator integration."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import aster_oracle as A  # noqa: E402
from tests.aster_common import fixture, layout, state_dict  # noqa: E402

DEV = "cuda"


def _build(max_len):
    from tpgsr_amd.model import recognizer
    voc = A.get_vocabulary("all")
    net = recognizer.RecognizerBuilder(arch="ResNet_ASTER", rec_num_classes=len(voc), sDim=512, attDim=512, max_len_labels=max_len,
                                       eos=voc.index("EOS"), STN_ON=True)
    return net, voc


def test_state_dict_layout_matches_reference():
    net, _ = _build(12)
    assert [(k, list(v.shape)) for k, v in net.state_dict().items()] == [(k, list(s)) for k, s in layout()]


def test_aster_greedy_matches_reference_fixture():
    from tpgsr_amd.interfaces.super_resolution import parse_aster_data
    g = fixture()
    net, voc = _build(int(g["max_len"]))
    net.load_state_dict(state_dict(int(g["seed"])), strict=True)
    net = net.to(DEV).eval()
    lr = torch.tensor(g["lr"]).to(DEV)
    d = parse_aster_data(lr)
    err = (d["images"].cpu() - torch.tensor(g["images"])).abs().max().item()
    print("parse_aster_data max err", err)
    assert err < 2e-6
    out = net(d)["output"]
    torch.cuda.synchronize()
    # rectified image: the TPS system is ill-conditioned in fp32: the reference's fp32 source coordinates and
    # this path's (fp64-accumulated) ones differ by ~1e-5 of the image width, which a noise image with O(1) pixel-to-pixel
    # differences turns into ~2e-3; control points agree to 1e-7 and the features / decisions downstream are unaffected
    for k, name, tol in (("ctrl", "ctrl_points", 2e-5), ("rectified", "rectified_images", 5e-3), ("feats", "encoder_feats", 1e-3)):
        e = (out[name].cpu() - torch.tensor(g[k])).abs().max().item()
        print(f"{k}: max abs err {e:.2e} (|ref| max {np.abs(g[k]).max():.2f})")
        assert e < tol, k
    ids, scores = out["pred_rec"].cpu(), out["pred_rec_score"].cpu()
    assert float(g["margin"].min()) > 2e-3            # every greedy decision of the fixture is clear
    assert torch.equal(ids, torch.tensor(g["ids"])), (ids.tolist(), g["ids"].tolist())
    assert (scores - torch.tensor(g["scores"])).abs().max().item() < 2e-3      # (classifier sharpened 40x in the fixture recipe)
    from tpgsr_amd.utils.metrics import get_string_aster
    assert get_string_aster(ids, voc) == A.get_string_aster(torch.tensor(g["ids"]), voc)


def test_aster_batch_vs_oracle():
    """a larger batch (N = 6) of fresh images against the oracle: features, and ids wherever the oracle's decision is clear"""
    g = fixture()
    net, voc = _build(10)
    sd = state_dict(int(g["seed"]))
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).eval()
    from tpgsr_amd.interfaces.super_resolution import parse_aster_data
    lr = torch.rand(6, 4, 16, 64, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        o = A.aster_greedy(sd, A.parse_aster_data(lr), len(voc), 10)
    out = net(parse_aster_data(lr.to(DEV)))["output"]
    torch.cuda.synchronize()
    assert (out["encoder_feats"].cpu() - o["feats"]).abs().max().item() < 2e-4
    ids = out["pred_rec"].cpu()
    # greedy decoding feeds its decisions back: compare up to (not including) the first step where the oracle's top-2 margin is tiny
    from make_golden_aster import margins
    with torch.no_grad():
        m = margins(A, sd, o["feats"], len(voc), 10)
    for n in range(6):
        clear = (m[n] > 1e-3).long().cumprod(0).sum().item()
        assert torch.equal(ids[n, :clear], o["ids"][n, :clear]), (n, clear, ids[n].tolist(), o["ids"][n].tolist())


def test_encoder_and_decoder_on_the_reference_rectified_image():
    """the stages after the (ill-conditioned) TPS in isolation: the reference's own rectified image through the encoder, and the
    reference's own features through the decoder, at the arithmetic's tolerance"""
    g = fixture()
    net, voc = _build(int(g["max_len"]))
    net.load_state_dict(state_dict(int(g["seed"])), strict=True)
    net = net.to(DEV).eval()
    feats = net.encoder(torch.tensor(g["rectified"]).to(DEV))
    e = (feats.cpu() - torch.tensor(g["feats"])).abs().max().item()
    print(f"encoder on the reference's rectified image: max abs err {e:.2e}")
    assert e < 2e-5
    ids, scores = net.decoder.sample([torch.tensor(g["feats"]).to(DEV), None, None])
    assert torch.equal(ids.cpu(), torch.tensor(g["ids"]))
    es = (scores.cpu() - torch.tensor(g["scores"])).abs().max().item()
    print(f"decoder on the reference's features: score max abs err {es:.2e}")
    assert es < 2e-5


def test_evaluator_with_aster_recognizer():
    """TextSREvaluator.recognize with an ASTER evaluation recognizer == oracle strings (images whose greedy decisions are clear)"""
    from tpgsr_amd.interfaces.super_resolution import TextSREvaluator
    g = fixture()
    net, voc = _build(int(g["max_len"]))
    net.load_state_dict(state_dict(int(g["seed"])), strict=True)
    net = net.to(DEV).eval()
    ev = TextSREvaluator([], [None], recognizer=net)
    strs = ev.recognize(torch.tensor(g["lr"]).to(DEV))
    assert strs == A.get_string_aster(torch.tensor(g["ids"]), voc), strs


def test_training_mode_is_rejected():
    net, _ = _build(4)
    net = net.to(DEV).train()
    with pytest.raises(RuntimeError):
        net({"images": torch.zeros(1, 3, 32, 128, device=DEV)})
