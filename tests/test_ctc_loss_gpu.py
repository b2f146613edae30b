"""GPU: the `--use_label` branch of the train loop (interfaces/super_resolution.py:40, :347-366): tpgsr_ctc_loss (csrc/crnn.hip) against
torch.nn.CTCLoss(blank=0, reduction='none') -- the object the reference calls -- through the fixture tests/golden/make_golden_ctc.py wrote
(per-sample values and the gradient of mean(ctc * weighted_tics) on random logits; a C3-shaped two-step trajectory with use_label AND
use_distill composed of the reference's own modules), in both logit layouts ([T][N][C] of the module API, [N][T][C] of the fused step),
with empty / one-character / cut-to-15 / out-of-alphabet labels."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _operands(g):
    from oracle import tpgsr_oracle as O
    lv, wm, wt = O.collate_labels([str(w) for w in g["words"]])
    assert np.array_equal(lv.numpy(), g["label_vecs"]) and np.array_equal(wm.numpy(), g["weighted_mask"]) and np.array_equal(wt.numpy(), g["weighted_tics"])
    lens = (lv.sum(1).squeeze(1) > 0).float().sum(1).to(torch.int32)
    off = torch.zeros_like(lens)
    off[1:] = torch.cumsum(lens, 0)[:-1]
    return wm.to(torch.int32).to(DEV), off.to(DEV), lens.to(DEV), wt.float().to(DEV), int(lens.max())


@pytest.mark.parametrize("layout", ["TNC", "NTC"])
def test_ctc_kernel_vs_torch_ctcloss(golden_dir, layout):
    from tpgsr_amd import kernels as K
    g = np.load(os.path.join(golden_dir, "ctc_loss.npz"))
    tg, off, lens, tics, mx = _operands(g)
    logits = torch.tensor(g["logits"])                     # [T][N][C]
    T, N, C = logits.shape
    if layout == "TNC":
        x, sn, st = logits.to(DEV).contiguous(), C, N * C
    else:
        x, sn, st = logits.permute(1, 0, 2).contiguous().to(DEV), T * C, C
    nll = torch.full((N,), float("nan"), device=DEV)
    d = torch.ones_like(x)                                   # accumulate on top of ones
    K.ctc_loss(x, sn, st, tg, off, lens, tics, N, T, C, 0, 1.0 / N, nll, d, True, mx)
    torch.cuda.synchronize()
    ref_nll, ref_d = torch.tensor(g["nll"]), torch.tensor(g["dlogits"])
    got_d = (d - 1).cpu()
    if layout == "NTC":
        got_d = got_d.permute(1, 0, 2)
    e_n = ((nll.cpu() - ref_nll).abs() / ref_nll.abs()).max().item()
    e_d = (got_d - ref_d).abs().max().item() / ref_d.abs().max().item()
    print(f"ctc {layout}: nll rel err {e_n:.2e}, gradient rel err {e_d:.2e}; weighted mean {(nll.cpu() * tics.cpu()).mean().item():.5f} (reference {float(g['loss_alone']):.5f})")
    assert e_n < 2e-6 and e_d < 4e-5
    assert got_d[:, np.asarray(g["weighted_tics"]) == 0].abs().max().item() == 0.0        # zero-weight samples take no gradient
    # value-only call, same numbers
    nll2 = torch.empty(N, device=DEV)
    K.ctc_loss(x, sn, st, tg, off, lens, None, N, T, C, 0, 0.0, nll2, None, False, mx)
    assert torch.equal(nll2, nll)


def test_train_step_with_use_label_vs_reference(golden_dir, golden_policy):
    from test_crnn_gpu import _c3_models
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    t = np.load(os.path.join(golden_dir, "ctc_loss.npz"))
    labels = (torch.tensor(t["label_vecs"]), torch.tensor(t["weighted_mask"]), torch.tensor(t["weighted_tics"]))
    srs, stus, teacher, *_ = _c3_models()
    ts = TPGSRTrainStep(srs, stus, teacher, stu_iter=1, use_label=True)
    lr, hr = torch.tensor(t["lr"]).to(DEV), torch.tensor(t["hr"]).to(DEV)
    with pytest.raises(ValueError, match="labels"):
        ts.step(lr, hr)
    loss = ts.step(lr, hr, labels=labels)
    gn = ts.opt.grad_norm(srs[0])
    print("C3 + ctc step0", golden_policy.name, loss.item(), t["loss"][0], gn.item(), t["gnorm"][0])
    assert abs(loss.item() - t["loss"][0]) < golden_policy.tol(3e-4) * t["loss"][0]
    assert abs(gn.item() - t["gnorm"][0]) < 3e-3 * t["gnorm"][0]
    assert (ts.last_p.cpu().permute(1, 0, 2).argmax(-1).numpy() == t["prior_argmax_step0"]).all()
    l1 = ts.step(lr, hr, labels=labels).item()
    print("C3 + ctc step1", l1, t["loss"][1])
    assert abs(l1 - t["loss"][1]) < 2e-2 * t["loss"][1]


def test_collate_labels_product_equals_oracle():
    from oracle import tpgsr_oracle as O
    from tpgsr_amd.data import AlignCollate
    words = ["Hotel", "a", "", "OPEN-24h", "supercalifragilistic", "!!", "x1", "aab", "fourteenletter", "fifteenletters1"]
    ac = AlignCollate.__new__(AlignCollate)
    ac.a2d = {ch: i for i, ch in enumerate(AlignCollate.D2A)}
    for a, b in zip(O.collate_labels(words), ac.encode(words)):
        assert torch.equal(a.float(), b.float())
