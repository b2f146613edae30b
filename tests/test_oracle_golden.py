"""CPU: the oracle (oracle/tpgsr_oracle.py) against the committed golden fixtures that were produced by the
genuine reference (tests/golden/make_golden.py).  Runs everywhere (no GPU, no /root/reference)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tpgsr_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _close(a, b, tol, what=""):
    a = a.detach() if isinstance(a, torch.Tensor) else a
    a, b = torch.as_tensor(np.asarray(a)), torch.as_tensor(np.asarray(b))
    err = (a.double() - b.double()).abs().max().item()
    scale = max(1.0, b.double().abs().max().item())
    assert err <= tol * scale, f"{what}: {err:.3e}"


def test_state_dict_layouts(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "state_dict_layouts.json")))
    specs = {
        "tsrn_stn_mask": O.tsrn_spec(STN=True, mask=True),
        "tsrn_nostn_nomask": O.tsrn_spec(STN=False, mask=False),
        "tsrn_tl_stn_mask": O.tsrn_spec(STN=True, mask=True, text_prior=True),
        "crnn": O.crnn_spec(), "srcnn": O.srcnn_spec(),
    }
    for k, spec in specs.items():
        assert [(n, list(s)) for n, s, _ in spec] == [(n, list(s)) for n, s in ref[k]], k
    assert sum(int(np.prod(s)) for n, s, kd in specs["tsrn_stn_mask"] if kd not in ("bn_rm", "bn_rv", "bn_nbt", "tps")) == 2681677
    assert sum(int(np.prod(s)) for n, s, kd in specs["tsrn_tl_stn_mask"] if kd not in ("bn_rm", "bn_rv", "bn_nbt", "tps")) == 3545869
    assert sum(int(np.prod(s)) for n, s, kd in specs["crnn"] if kd not in ("bn_rm", "bn_rv", "bn_nbt")) == 8331301


def test_tps_constants(golden_dir):
    g = _load(golden_dir, "tps_buffers.npz")
    b = O.tps_buffers(16, 64)
    _close(b["inverse_kernel"], g["inverse_kernel"], 1e-5)
    _close(b["target_coordinate_repr"], g["target_coordinate_repr"], 1e-6)
    _close(O.stn_identity_ctrl_points(), g["stn_fc2_bias"], 1e-7)


def test_losses(golden_dir):
    g = _load(golden_dir, "losses.npz")
    a = torch.tensor(g["a"], requires_grad=True)
    b = torch.tensor(g["b"])
    l = O.image_loss(a, b, True, (1, 1e-4))
    l.backward()
    _close(l.item(), g["image_loss"], 1e-6)
    _close(a.grad, g["image_loss_grad"], 1e-6)
    p = torch.tensor(g["p"], requires_grad=True)
    s = O.semantic_loss(p, torch.tensor(g["q"]))
    s.backward()
    _close(s.item(), g["semantic_loss"], 1e-6)
    _close(p.grad, g["semantic_loss_grad"], 1e-6)
    _close(O.calculate_psnr(a.detach().abs(), b), g["psnr"], 1e-6)
    _close(O.gradient_map(a.detach()[:, :3]), g["gradient_map"], 1e-7)


@pytest.mark.parametrize("explicit", [True, False])
def test_whole_tsrn(golden_dir, explicit):
    g = _load(golden_dir, "model_tsrn.npz")
    sd = O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True), 101, tps_hw=(16, 64))
    lr, hr = torch.tensor(g["lr"]), torch.tensor(g["hr"])
    p = O.as_params(sd)
    y = O.tsrn_forward(p, lr, training=True, stn=True, explicit_rnn=explicit)
    _close(y.detach(), g["y_train"], 5e-5)
    loss = O.image_loss(y, hr).mean() * 100
    _close(loss.item(), g["loss"], 1e-5)
    loss.backward()
    names = [str(n) for n in g["grad_names"]]
    gmax = g["grad_norms"].max()
    for n, ref_norm, head in zip(names, g["grad_norms"], g["grad_heads"]):
        got = p[n].grad
        assert abs(got.double().norm().item() - ref_norm) <= 2e-3 * max(ref_norm, 1e-3 * gmax), n
    y_eval = O.tsrn_forward(O.as_params(sd, False), lr, training=False)
    _close(y_eval, g["y_eval"], 5e-5)


def test_whole_tsrn_tl(golden_dir):
    g = _load(golden_dir, "model_tsrn_tl.npz")
    sd = O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 102, tps_hw=(16, 64))
    lr, prior = torch.tensor(g["lr"]), torch.tensor(g["extra0"])
    y = O.tsrn_forward(O.as_params(sd, False), lr, prior, training=True, stn=True, text_prior=True, explicit_rnn=True)
    _close(y, g["y_train"], 5e-5)
    y = O.tsrn_forward(O.as_params(sd, False), lr, prior, training=False, text_prior=True)
    _close(y, g["y_eval"], 5e-5)


@pytest.mark.parametrize("explicit", [True, False])
def test_whole_crnn(golden_dir, explicit):
    g = _load(golden_dir, "model_crnn.npz")
    sd = O.recipe_state_dict(O.crnn_spec(), 103)
    gray = O.parse_crnn_data(torch.tensor(g["hr"]))
    _close(gray, g["gray"], 1e-6)
    y = O.crnn_forward(O.as_params(sd, False), gray, training=True, explicit_rnn=explicit)
    _close(y, g["y_train"], 5e-5)
    y = O.crnn_forward(O.as_params(sd, False), gray, training=False, explicit_rnn=explicit)
    _close(y, g["y_eval"], 5e-5)


def test_srcnn_c1(golden_dir):
    g = _load(golden_dir, "model_srcnn.npz")
    sd = O.recipe_state_dict(O.srcnn_spec(), 104)
    _close(O.srcnn_forward(O.as_params(sd, False), torch.tensor(g["lr"])), g["y"], 2e-5)
    t = _load(golden_dir, "train_c1.npz")
    p = O.as_params(sd)
    opt = O.AdamState([p[k] for k in O.trainable_keys(p)])
    for step in range(2):
        r = O.srcnn_train_step(p, opt, torch.tensor(g["lr"]), torch.tensor(g["hr"]))
        _close(r["loss"], t["loss"][step], 1e-5)


def test_train_c2_nostn_trajectory(golden_dir):
    t = _load(golden_dir, "train_c2_nostn.npz")
    sd = O.recipe_state_dict(O.tsrn_spec(STN=False, mask=True), 201)
    p = O.as_params(sd)
    opt = O.AdamState([p[k] for k in O.trainable_keys(p)])
    lr, hr = torch.tensor(t["lr"]), torch.tensor(t["hr"])
    for step in range(3):
        r = O.tsrn_train_step(p, opt, lr, hr, stn=False)
        _close(r["loss"], t["loss"][step], 1e-5, f"step{step}")
        _close(r["grad_norm"], t["gnorm"][step], 1e-4, f"step{step}")


def test_train_c2_step0(golden_dir):
    t = _load(golden_dir, "train_c2.npz")
    sd = O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True), 201, tps_hw=(16, 64))
    p = O.as_params(sd)
    opt = O.AdamState([p[k] for k in O.trainable_keys(p)])
    r = O.tsrn_train_step(p, opt, torch.tensor(t["lr"]), torch.tensor(t["hr"]))
    _close(r["loss"], t["loss"][0], 1e-5)
    _close(r["grad_norm"], t["gnorm"][0], 1e-4)


def test_train_c3_step0(golden_dir):
    t = _load(golden_dir, "train_c3.npz")
    ps = O.as_params(O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 301, tps_hw=(16, 64)))
    pt = O.as_params(O.recipe_state_dict(O.crnn_spec(), 302), False)
    pu = O.as_params(O.recipe_state_dict(O.crnn_spec(), 303))
    opt = O.AdamState([ps[k] for k in O.trainable_keys(ps)] + [pu[k] for k in O.trainable_keys(pu)])
    r = O.tpgsr_train_step([ps], [pu], pt, opt, torch.tensor(t["lr"]), torch.tensor(t["hr"]), stu_iter=1)
    _close(r["loss"], t["loss"][0], 1e-5)
    _close(r["grad_norms"][0], t["gnorm"][0], 1e-4)
    assert (r["priors"][0].argmax(-1).numpy() == t["prior_argmax_step0"]).all()


def test_explicit_rnn_equals_aten():
    g = torch.Generator().manual_seed(3)
    sd = O.recipe_state_dict(O._gru_spec("g", 64, 32), 7)
    x = torch.randn(5, 9, 64, generator=g)
    _close(O.gru_bidir(x, sd, "g", True), O.gru_bidir(x, sd, "g", False), 1e-6)
    sd = O.recipe_state_dict(O._lstm_spec("l", 24, 16), 8)
    x = torch.randn(7, 3, 24, generator=g)
    _close(O.lstm_bidir(x, sd, "l", True), O.lstm_bidir(x, sd, "l", False), 1e-6)


def test_eval_metrics_oracle_vs_reference_fixture(golden_dir):
    """oracle.get_string_crnn / ssim / calculate_psnr (restated from utils/metrics.py:71-88, utils/ssim_psnr.py) against the
    reference's own outputs (tests/golden/make_golden_next.py)"""
    g = np.load(os.path.join(golden_dir, "next_eval_metrics.npz"))
    assert O.get_string_crnn(torch.tensor(g["logits"])) == [str(s) for s in g["strings"]]
    a, b = torch.tensor(g["a"]), torch.tensor(g["b"])
    assert abs(float(O.calculate_psnr(a, b)) - float(g["psnr"])) < 1e-5
    assert abs(float(O.ssim(a, b)) - float(g["ssim"])) < 1e-6


def test_ssim_loss_branch_oracle_vs_reference_fixture(golden_dir):
    """`--ssim_loss` (interfaces/super_resolution.py:388-391): the oracle's SSIM, its gradient and the C3-shaped step with the SSIM term
    against tests/golden/ssim_loss.npz (written by make_golden_ssim.py from the imported reference)"""
    import os
    import numpy as np
    import torch
    from oracle import tpgsr_oracle as O
    g = np.load(os.path.join(golden_dir, "ssim_loss.npz"))
    for case in ("noise", "near"):
        x = torch.tensor(g[f"{case}_x"]).requires_grad_(True)
        v = O.ssim(x, torch.tensor(g[f"{case}_y"])).mean()
        (gr,) = torch.autograd.grad((1 - v) * 10., x)
        assert abs(v.item() - float(g[f"{case}_value"])) < 1e-6
        assert (gr - torch.tensor(g[f"{case}_grad"])).abs().max() < 1e-7
    lr, hr = torch.tensor(g["lr"]), torch.tensor(g["hr"])
    ps = O.as_params(O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 301, tps_hw=(16, 64)))
    pt, pu = O.as_params(O.recipe_state_dict(O.crnn_spec(), 302), False), O.as_params(O.recipe_state_dict(O.crnn_spec(), 303))
    opt = O.AdamState([ps[k] for k in O.trainable_keys(ps)] + [pu[k] for k in O.trainable_keys(pu)])
    r = O.tpgsr_train_step([ps], [pu], pt, opt, lr, hr, stu_iter=1, ssim_loss=True)
    assert abs(float(r["loss"]) - g["loss"][0]) < 1e-4 * g["loss"][0]
    assert abs(float(r["grad_norms"][0]) - g["gnorm"][0]) < 1e-3 * g["gnorm"][0]
    assert (r["priors"][0].argmax(-1).numpy() == g["prior_argmax_step0"]).all()


def test_use_label_branch_oracle_vs_reference_fixture(golden_dir):
    """`--use_label` (interfaces/super_resolution.py:347-366): the oracle's collate labels, CTC term and C3-shaped step against
    tests/golden/ctc_loss.npz (written by make_golden_ctc.py: reference modules + torch.nn.CTCLoss as the reference file calls it)"""
    import os
    import numpy as np
    import torch
    import torch.nn.functional as F
    from oracle import tpgsr_oracle as O
    g = np.load(os.path.join(golden_dir, "ctc_loss.npz"))
    lv, wm, wt = O.collate_labels([str(w) for w in g["words"]])
    assert np.array_equal(lv.numpy(), g["label_vecs"]) and np.array_equal(wm.numpy(), g["weighted_mask"]) and np.array_equal(wt.numpy(), g["weighted_tics"])
    logits = torch.tensor(g["logits"])
    tl = (lv.sum(1).squeeze(1) > 0).float().sum(1).long()
    nll = F.ctc_loss(logits.log_softmax(2), wm, torch.full((logits.shape[1],), logits.shape[0], dtype=torch.long), tl, blank=0, reduction="none")
    assert (nll - torch.tensor(g["nll"])).abs().max() < 1e-4
    lr, hr = torch.tensor(g["lr"]), torch.tensor(g["hr"])
    ps = O.as_params(O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 301, tps_hw=(16, 64)))
    pt, pu = O.as_params(O.recipe_state_dict(O.crnn_spec(), 302), False), O.as_params(O.recipe_state_dict(O.crnn_spec(), 303))
    opt = O.AdamState([ps[k] for k in O.trainable_keys(ps)] + [pu[k] for k in O.trainable_keys(pu)])
    r = O.tpgsr_train_step([ps], [pu], pt, opt, lr, hr, stu_iter=1, use_label=True, labels=(lv, wm, wt))
    assert abs(float(r["loss"]) - g["loss"][0]) < 1e-4 * g["loss"][0]
    assert abs(float(r["grad_norms"][0]) - g["gnorm"][0]) < 1e-3 * g["gnorm"][0]
