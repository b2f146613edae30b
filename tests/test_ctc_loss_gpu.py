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


def test_cascade_with_ssim_loss_and_labels_vs_oracle():
    """Both optional loss branches at once in a TWO-stage cascade (stu_iter 2, sr_share, two students, no STN, bs 4): the CTC term of
    every stage's student and the SSIM term of every stage's SR image, loss value and raw gradients against oracle autograd
    (oracle/tpgsr_oracle.py: tpgsr_train_step(ssim_loss=True, use_label=True), pinned at stu_iter 1 by the two reference fixtures).
    Tolerances as tests/test_crnn_gpu.py::test_cascade_two_stages_vs_oracle (the students' gradients are percent-level in fp32)."""
    from oracle import tpgsr_oracle as O
    from test_crnn_gpu import _c3_models
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    srs, stus, teacher, sds, sd_s, sd_t = _c3_models(stn=False, n_sr=1, n_stu=2)
    lr, hr = O.synthetic_batch(4, 78)
    labels = O.collate_labels(["Hotel", "a1", "", "OPEN-24h"])
    ps = O.as_params(sds[0]); pt = O.as_params(sd_t, False); pu = [O.as_params(x) for x in sd_s]
    opt = O.AdamState([ps[k] for k in O.trainable_keys(ps)] + [q[k] for q in pu for k in O.trainable_keys(q)])
    ref = O.tpgsr_train_step([ps], pu, pt, opt, lr, hr, stu_iter=2, sr_share=True, tpg_share=False, stn=False, ssim_loss=True,
                             use_label=True, labels=labels)
    ts = TPGSRTrainStep(srs, stus, teacher, stu_iter=2, sr_share=True, tpg_share=False, ssim_loss=True, use_label=True)
    ts.pool.bind(torch.device(DEV, 0))
    teacher._engine().bind(torch.device(DEV, 0))
    loss = ts._phase_a(lr.to(DEV), hr.to(DEV), labels)            # forward + backward only (no optimiser): raw gradients
    torch.cuda.synchronize()
    print("cascade + ssim + ctc: loss", loss.item(), ref["loss"].item())
    assert abs(loss.item() - ref["loss"].item()) < 3e-4 * ref["loss"].item()
    flat_ref = ref["grads"]
    names_sr = O.trainable_keys(ps)
    n_sr = len(names_sr)
    coef = min(1.0, 0.25 / (float(ref["grad_norms"][0]) + 1e-6))   # the oracle clipped the SR group in place
    P = dict(srs[0].named_parameters())
    num = den = 0.0
    for k, gref in zip(names_sr, flat_ref[:n_sr]):
        d = P[k].grad.cpu() * coef - gref
        num += d.double().pow(2).sum().item(); den += gref.double().pow(2).sum().item()
    print("  SR gradients: global rel err", (num / den) ** 0.5)
    assert (num / den) ** 0.5 < 5e-3
    ofs = n_sr
    for j, q in enumerate(pu):
        Ps = dict(stus[j].named_parameters())
        keys = O.trainable_keys(q)
        num = den = 0.0
        for k, gref in zip(keys, flat_ref[ofs:ofs + len(keys)]):
            d = Ps[k].grad.cpu() - gref
            num += d.double().pow(2).sum().item(); den += gref.double().pow(2).sum().item()
        print(f"  student {j}: global rel err {(num / den) ** 0.5:.3e}")
        assert (num / den) ** 0.5 < 3e-2, (j, (num / den) ** 0.5)
        ofs += len(keys)


def test_module_api_loop_with_use_label(golden_dir):
    """the reference's loop body with `--use_label` on the drop-in modules (interfaces/super_resolution.py:347-366): torch's OWN
    `nn.CTCLoss(blank=0, reduction='none')` on the drop-in student's logits, autograd through it into the HIP backward plans --
    the line keeps working as it is."""
    from test_crnn_gpu import _c3_models
    from tpgsr_amd.interfaces.super_resolution import parse_crnn_data
    from tpgsr_amd.loss.image_loss import ImageLoss
    from tpgsr_amd.loss.semantic_loss import SemanticLoss
    t = np.load(os.path.join(golden_dir, "ctc_loss.npz"))
    srs, stus, teacher, *_ = _c3_models()
    model, stu = srs[0], stus[0]
    for q in teacher.parameters():
        q.requires_grad = False
    image_crit, sem_loss = ImageLoss(gradient=True, loss_weight=[1, 1e-4]), SemanticLoss()
    ctc_loss = torch.nn.CTCLoss(blank=0, reduction='none')
    optimizer_G = torch.optim.Adam(list(model.parameters()) + list(stu.parameters()), lr=1e-3, betas=(0.5, 0.999))
    images_lr, images_hr = torch.tensor(t["lr"]).to(DEV), torch.tensor(t["hr"]).to(DEV)
    label_vecs_gt, weighted_mask, weighted_tics = torch.tensor(t["label_vecs"]), torch.tensor(t["weighted_mask"]), torch.tensor(t["weighted_tics"])
    text_sum = label_vecs_gt.sum(1).squeeze(1)
    text_len = (text_sum > 0).float().sum(1).reshape(-1)
    label_vecs_hr = torch.nn.functional.softmax(teacher(parse_crnn_data(images_hr[:, :3, :, :])).detach(), -1)
    label_vecs_logits = stu(parse_crnn_data(images_lr[:, :3, :, :]))
    label_vecs = torch.nn.functional.softmax(label_vecs_logits, -1)
    label_vecs_final = label_vecs.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
    predicted_length = torch.ones(label_vecs_logits.shape[1]) * label_vecs_logits.shape[0]
    fsup_sem_loss = ctc_loss(label_vecs_logits.log_softmax(2), weighted_mask.long().to(DEV), predicted_length.long().to(DEV), text_len.long().to(DEV))
    loss_recog_distill = (fsup_sem_loss * weighted_tics.float().to(DEV)).mean() + sem_loss(label_vecs, label_vecs_hr) * 100
    drop_vec = torch.ones(images_lr.shape[0]).float()
    drop_vec[:int(images_lr.shape[0] // 4)] = 0.
    label_vecs_final = label_vecs_final * drop_vec.to(DEV).view(-1, 1, 1, 1)
    cascade_images = model(images_lr, label_vecs_final)
    loss_img = image_crit(cascade_images, images_hr).mean() * 100
    loss_im = loss_img + loss_recog_distill
    optimizer_G.zero_grad()
    loss_im.backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25)
    l_c = float((fsup_sem_loss.detach() * weighted_tics.float().to(DEV)).mean())
    print("module API + ctc:", loss_im.item(), t["loss"][0], l_c, t["loss_ctc"][0], float(gn), t["gnorm"][0])
    assert abs(l_c - t["loss_ctc"][0]) < 1e-4 * abs(t["loss_ctc"][0])
    assert abs(loss_im.item() - t["loss"][0]) < 3e-4 * t["loss"][0]
    assert abs(float(gn) - t["gnorm"][0]) < 3e-3 * t["gnorm"][0]
