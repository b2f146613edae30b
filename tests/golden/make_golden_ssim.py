#!/usr/bin/env python
"""BUILD CONTAINER ONLY: fixtures of the `--ssim_loss` branch of the train loop (interfaces/super_resolution.py:388-391:
`loss_ssim = (1 - ssim(cascade_images, images_hr).mean()) * 10.; loss_img += loss_ssim`, ssim = utils.ssim_psnr.SSIM()).
Imports the genuine reference (oracle/ref_import.py), hard-asserts that the oracle's restatement equals it -- the SSIM value and its
gradient with respect to the first image on noise images and on an SR-like pair; a C3-shaped two-step trajectory composed of the
reference's OWN modules with the SSIM term in the loss -- and writes tests/golden/ssim_loss.npz (data only)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import, tpgsr_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def close(a, b, tol, what):
    a, b = float(a), float(b)
    assert abs(a - b) <= tol * max(1.0, abs(b)), f"{what}: oracle {a} vs reference {b}"


def main():
    R = ref_import.load()
    torch.manual_seed(0)
    ssim_ref = R.ssim_psnr.SSIM()
    # 1. the function and its gradient
    g = torch.Generator().manual_seed(41)
    a = torch.rand(3, 4, 32, 128, generator=g)
    b = (a + 0.1 * torch.randn(3, 4, 32, 128, generator=g)).clamp(0, 1)
    cases = {}
    for name, (x, y) in {"noise": (torch.rand(3, 4, 32, 128, generator=g), torch.rand(3, 4, 32, 128, generator=g)), "near": (b, a)}.items():
        xr = x.clone().requires_grad_(True)
        v_ref = ssim_ref(xr, y).mean()
        (gr,) = torch.autograd.grad((1 - v_ref) * 10., xr)
        xo = x.clone().requires_grad_(True)
        v_or = O.ssim(xo, y).mean()
        (go,) = torch.autograd.grad((1 - v_or) * 10., xo)
        close(v_or, v_ref, 1e-6, f"ssim {name}")
        assert (go - gr).abs().max() <= 1e-6 * max(1.0, gr.abs().max().item()), f"ssim gradient {name}"
        assert gr[:, 3:].abs().max() == 0                     # the mask channel takes no SSIM gradient
        cases[name] = (x, y, v_ref.detach(), gr)
        print(f"  ssim {name}: {float(v_ref):.6f}, |grad| max {float(gr.abs().max()):.3e}: oracle == reference")
    # 2. C3-shaped trajectory with the SSIM term, from the reference's own modules
    lr, hr = O.synthetic_batch(4, 7)
    sd_sr = O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 301, tps_hw=(16, 64))
    sd_t, sd_s = O.recipe_state_dict(O.crnn_spec(), 302), O.recipe_state_dict(O.crnn_spec(), 303)
    net = R.tsrn.TSRN_TL(STN=True, mask=True); net.load_state_dict(sd_sr); net.train()
    teacher = R.crnn.CRNN(32, 1, 37, 256); teacher.load_state_dict(sd_t); teacher.eval()
    for q in teacher.parameters():
        q.requires_grad = False
    stu = R.crnn.CRNN(32, 1, 37, 256); stu.load_state_dict(sd_s); stu.train()
    opt = torch.optim.Adam(list(net.parameters()) + list(stu.parameters()), lr=1e-3, betas=(0.5, 0.999))
    sem, crit = R.semantic_loss.SemanticLoss(), R.image_loss.ImageLoss(gradient=True, loss_weight=[1, 1e-4])
    ps, pt, pu = O.as_params(sd_sr), O.as_params(sd_t, False), O.as_params(sd_s)
    oopt = O.AdamState([ps[k] for k in O.trainable_keys(ps)] + [pu[k] for k in O.trainable_keys(pu)])
    traj = {"loss": [], "gnorm": [], "loss_ssim": []}
    for step in range(2):
        hr_prior = F.softmax(teacher(O.parse_crnn_data(hr[:, :3])).detach(), -1)
        logits = stu(O.parse_crnn_data(lr[:, :3]))
        pv = F.softmax(logits, -1)
        pf = pv.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
        l_d = sem(pv, hr_prior) * 100
        drop = torch.ones(4); drop[:1] = 0
        pf = pf * drop.view(-1, 1, 1, 1)
        sr = net(lr, pf)
        l_i = crit(sr, hr).mean() * 100
        l_s = (1 - ssim_ref(sr, hr).mean()) * 10.          # super_resolution.py:390
        loss = l_i + l_s + l_d
        opt.zero_grad(); loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 0.25)
        opt.step()
        r = O.tpgsr_train_step([ps], [pu], pt, oopt, lr, hr, stu_iter=1, ssim_loss=True)
        ltol = (1e-4, 5e-4)[step]
        close(r["loss"], loss, ltol, f"C3+ssim step{step} loss"); close(r["grad_norms"][0], gn, 10 * ltol, f"C3+ssim step{step} gnorm")
        if step == 0:
            prior_argmax = pv.detach().argmax(-1).numpy()
            assert (r["priors"][0].argmax(-1).numpy() == prior_argmax).all()
        traj["loss"].append(loss.item()); traj["gnorm"].append(float(gn)); traj["loss_ssim"].append(l_s.item())
        print(f"  C3+ssim step {step}: loss {loss.item():.6f} (img {l_i.item():.5f} ssim {l_s.item():.5f} distill {l_d.item():.5f}) gnorm {float(gn):.5f}")
    np.savez_compressed(os.path.join(OUT, "ssim_loss.npz"),
                        **{f"{n}_{k}": v.numpy() for n, (x, y, val, gr) in cases.items() for k, v in (("x", x), ("y", y), ("value", val), ("grad", gr))},
                        lr=lr.numpy(), hr=hr.numpy(), loss=np.array(traj["loss"]), gnorm=np.array(traj["gnorm"]), loss_ssim=np.array(traj["loss_ssim"]),
                        prior_argmax_step0=prior_argmax)
    print("ssim_loss.npz written")


if __name__ == "__main__":
    main()
