"""GPU: the `--ssim_loss` branch of the train loop (interfaces/super_resolution.py:388-391: loss_img += (1 - ssim(cascade_images, images_hr).mean())
* 10.) -- the differentiable SSIM module (tpgsr_ssim / tpgsr_ssim_bwd, csrc/metrics.hip) and TPGSRTrainStep(ssim_loss=True) against the
fixture tests/golden/make_golden_ssim.py wrote from the imported reference (utils/ssim_psnr.py:30-78 under autograd; a C3-shaped two-step
trajectory composed of the reference's own modules)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("case", ["noise", "near"])
def test_ssim_module_value_and_gradient_vs_reference(golden_dir, case):
    from tpgsr_amd.utils.ssim_psnr import SSIM
    g = np.load(os.path.join(golden_dir, "ssim_loss.npz"))
    x = torch.tensor(g[f"{case}_x"]).to(DEV).requires_grad_(True)
    y = torch.tensor(g[f"{case}_y"]).to(DEV)
    v = SSIM()(x, y).mean()
    loss = (1 - v) * 10.
    loss.backward()
    ref_v, ref_g = float(g[f"{case}_value"]), torch.tensor(g[f"{case}_grad"])
    e_v = abs(v.item() - ref_v)
    e_g = (x.grad.cpu() - ref_g).abs().max().item() / ref_g.abs().max().item()
    print(f"ssim {case}: value {v.item():.7f} (reference {ref_v:.7f}), gradient rel err {e_g:.2e}")
    assert e_v < 2e-6 and e_g < 2e-5
    assert x.grad[:, 3:].abs().max().item() == 0.0          # the mask channel takes no SSIM gradient (img1[:, :3])
    # without a gradient request the module is the evaluation metric it always was
    with torch.no_grad():
        assert abs(SSIM()(x.detach(), y).item() - ref_v) < 2e-6


def test_train_step_with_ssim_loss_vs_reference(golden_dir, golden_policy):
    from test_crnn_gpu import _c3_models
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    t = np.load(os.path.join(golden_dir, "ssim_loss.npz"))
    srs, stus, teacher, *_ = _c3_models()
    ts = TPGSRTrainStep(srs, stus, teacher, stu_iter=1, ssim_loss=True)
    lr, hr = torch.tensor(t["lr"]).to(DEV), torch.tensor(t["hr"]).to(DEV)
    loss = ts.step(lr, hr)
    gn = ts.opt.grad_norm(srs[0])
    print("C3 + ssim step0", golden_policy.name, loss.item(), t["loss"][0], gn.item(), t["gnorm"][0])
    assert abs(loss.item() - t["loss"][0]) < golden_policy.tol(3e-4) * t["loss"][0]
    assert abs(gn.item() - t["gnorm"][0]) < 3e-3 * t["gnorm"][0]
    assert (ts.last_p.cpu().permute(1, 0, 2).argmax(-1).numpy() == t["prior_argmax_step0"]).all()
    l1 = ts.step(lr, hr).item()
    print("C3 + ssim step1", l1, t["loss"][1])
    assert abs(l1 - t["loss"][1]) < 2e-2 * t["loss"][1]


def test_module_api_loop_with_ssim_loss(golden_dir):
    """the reference's loop body with `--ssim_loss` on the drop-in modules: autograd runs through tpgsr_amd.utils.ssim_psnr.SSIM"""
    from test_crnn_gpu import _c3_models
    from tpgsr_amd.interfaces.super_resolution import parse_crnn_data
    from tpgsr_amd.loss.image_loss import ImageLoss
    from tpgsr_amd.loss.semantic_loss import SemanticLoss
    from tpgsr_amd.utils import ssim_psnr
    t = np.load(os.path.join(golden_dir, "ssim_loss.npz"))
    srs, stus, teacher, *_ = _c3_models()
    model, stu = srs[0], stus[0]
    for q in teacher.parameters():
        q.requires_grad = False
    image_crit, sem_loss, ssim = ImageLoss(gradient=True, loss_weight=[1, 1e-4]), SemanticLoss(), ssim_psnr.SSIM()
    optimizer_G = torch.optim.Adam(list(model.parameters()) + list(stu.parameters()), lr=1e-3, betas=(0.5, 0.999))
    images_lr, images_hr = torch.tensor(t["lr"]).to(DEV), torch.tensor(t["hr"]).to(DEV)
    label_vecs_hr = torch.nn.functional.softmax(teacher(parse_crnn_data(images_hr[:, :3, :, :])).detach(), -1)
    label_vecs = torch.nn.functional.softmax(stu(parse_crnn_data(images_lr[:, :3, :, :])), -1)
    label_vecs_final = label_vecs.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
    loss_recog_distill = sem_loss(label_vecs, label_vecs_hr) * 100
    drop_vec = torch.ones(images_lr.shape[0]).float()
    drop_vec[:int(images_lr.shape[0] // 4)] = 0.
    label_vecs_final = label_vecs_final * drop_vec.to(DEV).view(-1, 1, 1, 1)
    cascade_images = model(images_lr, label_vecs_final)
    loss_img = image_crit(cascade_images, images_hr).mean() * 100
    loss_ssim = (1 - ssim(cascade_images, images_hr).mean()) * 10.
    loss_im = loss_img + loss_ssim + loss_recog_distill
    optimizer_G.zero_grad()
    loss_im.backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25)
    print("module API + ssim:", loss_im.item(), t["loss"][0], float(loss_ssim.detach()), t["loss_ssim"][0], float(gn), t["gnorm"][0])
    assert abs(float(loss_ssim.detach()) - t["loss_ssim"][0]) < 1e-4 * t["loss_ssim"][0]
    assert abs(loss_im.item() - t["loss"][0]) < 3e-4 * t["loss"][0]
    assert abs(float(gn) - t["gnorm"][0]) < 3e-3 * t["gnorm"][0]
