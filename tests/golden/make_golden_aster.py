#!/usr/bin/env python
"""Fixture for SURVEY.md section 8 row N2, ASTER evaluation recognizer (greedy decode), generated from the GENUINE reference imported
from /root/reference (build container only; stubs for IPython / torchvision, `Tensor.cuda` = identity because the reference calls
.cuda() unconditionally).  Asserts oracle/aster_oracle.py == reference stage by stage, then stores inputs + expected outputs only
(weights by the same `generic_recipe` as make_golden_next.py).

    python tests/golden/make_golden_aster.py        # rewrites tests/golden/aster_eval.npz + aster_layout.json"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden_next import generic_recipe  # noqa: E402

MAX_LEN = 12       # the reference evaluates with max_len 100; the recurrence is identical, 12 steps keep the fixture small


def weights(template, seed):
    """generic recipe + a sharper classifier and perturbed control points, so that arg-max decisions and the rectification are
    non-trivial (the tests rebuild exactly this from the seed stored in the fixture)"""
    from oracle import tpgsr_oracle as O
    sd = generic_recipe(template, seed)
    rng = np.random.default_rng(seed + 1)
    sd["decoder.decoder.fc.weight"] = sd["decoder.decoder.fc.weight"] * 40.0
    sd["stn_head.stn_fc2.weight"] = torch.tensor(rng.normal(0, 0.02, tuple(sd["stn_head.stn_fc2.weight"].shape)), dtype=torch.float32)
    sd["stn_head.stn_fc2.bias"] = O.stn_identity_ctrl_points(20, 0.01).reshape(-1) + torch.tensor(rng.normal(0, 0.01, 40), dtype=torch.float32)
    for k, v in O.tps_buffers(32, 100, 20, (0.05, 0.05)).items():      # the TPS constants are buffers of every checkpoint, not weights
        sd["tps." + k] = v.clone()
    return sd


def margins(A, p, feats, ncls, max_len=None):
    """top-2 probability margin of every greedy decision"""
    xproj = torch.nn.functional.linear(feats, p["decoder.decoder.attention_unit.xEmbed.weight"], p["decoder.decoder.attention_unit.xEmbed.bias"])
    s = feats.new_zeros(feats.shape[0], 512)
    y = torch.full((feats.shape[0],), ncls, dtype=torch.long)
    out = []
    for _ in range(max_len or MAX_LEN):
        logits, s = A.decoder_step(p, "decoder.decoder", feats, xproj, s, y)
        pr = torch.softmax(logits, 1)
        t2 = pr.topk(2, 1).values
        out.append(t2[:, 0] - t2[:, 1])
        y = pr.argmax(1)
    return torch.stack(out, 1)


def main():
    for name in ("IPython", "cv2"):
        m = types.ModuleType(name)
        m.embed = lambda *a, **k: None
        sys.modules.setdefault(name, m)
    tv = types.ModuleType("torchvision")
    for sub in ("models", "transforms", "datasets"):
        m = types.ModuleType("torchvision." + sub)
        setattr(tv, sub, m)
        sys.modules["torchvision." + sub] = m
    sys.modules.setdefault("torchvision", tv)
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, "/root/reference")
    import warnings
    warnings.filterwarnings("ignore")
    from model.recognizer.recognizer_builder import RecognizerBuilder
    from oracle import aster_oracle as A

    voc = A.get_vocabulary("all")
    from utils import labelmaps
    assert voc == labelmaps.get_vocabulary("all", EOS="EOS", PADDING="PADDING", UNKNOWN="UNKNOWN")
    ncls = len(voc)
    torch.manual_seed(0)
    ref = RecognizerBuilder(arch="ResNet_ASTER", rec_num_classes=ncls, sDim=512, attDim=512, max_len_labels=MAX_LEN,
                            eos=voc.index("EOS"), STN_ON=True)
    layout = [(k, list(v.shape)) for k, v in ref.state_dict().items()]
    for k, v in A.O.tps_buffers(32, 100, 20, (0.05, 0.05)).items():       # the oracle's TPS constants == the reference module's
        assert (ref.state_dict()["tps." + k] - v).abs().max().item() < 1e-6, k
    g = torch.Generator().manual_seed(31)
    lr = torch.rand(2, 4, 16, 64, generator=g)
    for seed in range(4242, 4342):      # first recipe seed whose greedy decisions are all clear (top-2 margin > 2e-3)
        sd = weights(ref.state_dict(), seed)
        ref.load_state_dict(sd, strict=True)
        ref.eval()
        with torch.no_grad():
            images = torch.nn.functional.interpolate(lr[:, :3], (32, 128), mode="bicubic") * 2 - 1      # parse_aster_data
            assert torch.equal(images, A.parse_aster_data(lr))
            stn_in = torch.nn.functional.interpolate(images, [32, 64], mode="bilinear", align_corners=True)
            _, ctrl = ref.stn_head(stn_in)
            rect, _ = ref.tps(images, ctrl)
            feats = ref.encoder(rect).contiguous()
            ids, scores = ref.decoder.sample([feats, None, None])
            p = {k: v for k, v in ref.state_dict().items()}
            margin = margins(A, p, feats, ncls)
        if float(margin.min()) > 2e-3 and len(set(ids.flatten().tolist())) >= 4:
            break
    else:
        raise SystemExit("no seed with clear greedy decisions")
    with torch.no_grad():
        o = A.aster_greedy(p, images, ncls, MAX_LEN)
    for k, r in (("ctrl", ctrl), ("rectified", rect), ("feats", feats), ("scores", scores)):
        err = (o[k] - r).abs().max().item()
        print(f"oracle vs reference {k}: max abs diff {err:.3e}")
        assert err < 2e-5, k
    assert torch.equal(o["ids"], ids)
    print("greedy ids:", ids.tolist(), "min top-2 margin", float(margin.min()))
    print("strings:", A.get_string_aster(ids, voc))
    np.savez_compressed(os.path.join(HERE, "aster_eval.npz"), lr=lr.numpy(), images=images.numpy(), ctrl=ctrl.numpy(),
                        rectified=rect.numpy(), feats=feats.numpy(), ids=ids.numpy(), scores=scores.numpy(), margin=margin.numpy(),
                        max_len=np.array(MAX_LEN), seed=np.array(seed))
    json.dump({"aster": layout}, open(os.path.join(HERE, "aster_layout.json"), "w"))
    print("wrote aster_eval.npz, aster_layout.json;", len(layout), "state_dict entries")


if __name__ == "__main__":
    main()
