#!/usr/bin/env python
"""BUILD CONTAINER ONLY: fixtures of the `--use_label` branch of the train loop (interfaces/super_resolution.py:40, :347-366:
`ctc_loss = torch.nn.CTCLoss(blank=0, reduction='none')` on `label_vecs_logits.log_softmax(2)`, targets = the collate's concatenated
`weighted_mask`, lengths derived from the one-hot `label_vecs`, per-sample weights `weighted_tics`, `.mean()`).
Imports the genuine reference's modules (oracle/ref_import.py) for the networks, composes the loop body exactly as the reference file does
-- the CTC loss object is torch's own, as in the reference -- hard-asserts oracle == that composition and writes tests/golden/ctc_loss.npz.
The label tensors come from oracle.collate_labels, the restatement of dataset/dataset.py:1255-1323 (dataset/* cannot be imported: cv2,
lmdb, pyfasttext are absent -- SURVEY 8c)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import, tpgsr_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
WORDS = ["Hotel", "a", "", "OPEN-24h", "supercalifragilistic", "!!", "x1", "aab"]


def close(a, b, tol, what):
    a, b = float(a), float(b)
    assert abs(a - b) <= tol * max(1.0, abs(b)), f"{what}: oracle {a} vs reference {b}"


def main():
    R = ref_import.load()
    torch.manual_seed(0)
    ctc_loss = torch.nn.CTCLoss(blank=0, reduction='none')          # interfaces/super_resolution.py:40
    # 1. the loss alone on random logits (T 26, N 8, C 37): per-sample values and the gradient of mean(ctc * tics)
    g = torch.Generator().manual_seed(5)
    label_vecs, weighted_mask, weighted_tics = O.collate_labels(WORDS)
    logits = (torch.randn(26, 8, 37, generator=g) * 2).requires_grad_(True)
    text_sum = label_vecs.sum(1).squeeze(1)
    text_len = (text_sum > 0).float().sum(1).reshape(-1)
    predicted_length = torch.ones(logits.shape[1]) * logits.shape[0]
    fsup = ctc_loss(logits.log_softmax(2), weighted_mask.long(), predicted_length.long(), text_len.long())
    loss = (fsup * weighted_tics.float()).mean()
    (dlog,) = torch.autograd.grad(loss, logits)
    print("  per-sample CTC:", [round(float(v), 4) for v in fsup], "mean(weighted)", float(loss))
    assert torch.isfinite(fsup).all()
    # 2. C3-shaped two-step trajectory with use_label AND use_distill (both branches of :347-373), reference modules
    lr, hr = O.synthetic_batch(8, 9)
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
    traj = {"loss": [], "gnorm": [], "loss_ctc": []}
    for step in range(2):
        hr_prior = F.softmax(teacher(O.parse_crnn_data(hr[:, :3])).detach(), -1)
        label_vecs_logits = stu(O.parse_crnn_data(lr[:, :3]))
        pv = F.softmax(label_vecs_logits, -1)
        pf = pv.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
        predicted_length = torch.ones(label_vecs_logits.shape[1]) * label_vecs_logits.shape[0]
        fsup_sem_loss = ctc_loss(label_vecs_logits.log_softmax(2), weighted_mask.long(), predicted_length.long(), text_len.long())
        l_c = (fsup_sem_loss * weighted_tics.float()).mean()
        l_d = sem(pv, hr_prior) * 100
        drop = torch.ones(8); drop[:2] = 0
        pf = pf * drop.view(-1, 1, 1, 1)
        sr = net(lr, pf)
        l_i = crit(sr, hr).mean() * 100
        loss_im = l_i + l_c + l_d
        opt.zero_grad(); loss_im.backward()
        gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 0.25)
        opt.step()
        r = O.tpgsr_train_step([ps], [pu], pt, oopt, lr, hr, stu_iter=1, use_label=True, labels=(label_vecs, weighted_mask, weighted_tics))
        ltol = (1e-4, 5e-4)[step]
        close(r["loss"], loss_im, ltol, f"C3+ctc step{step} loss"); close(r["grad_norms"][0], gn, 10 * ltol, f"C3+ctc step{step} gnorm")
        if step == 0:
            prior_argmax = pv.detach().argmax(-1).numpy()
            assert (r["priors"][0].argmax(-1).numpy() == prior_argmax).all()
        traj["loss"].append(loss_im.item()); traj["gnorm"].append(float(gn)); traj["loss_ctc"].append(l_c.item())
        print(f"  C3+ctc step {step}: loss {loss_im.item():.6f} (img {l_i.item():.5f} ctc {l_c.item():.5f} distill {l_d.item():.5f}) gnorm {float(gn):.5f}")
    np.savez_compressed(os.path.join(OUT, "ctc_loss.npz"), words=np.array(WORDS), label_vecs=label_vecs.numpy(), weighted_mask=weighted_mask.numpy(),
                        weighted_tics=weighted_tics.numpy(), logits=logits.detach().numpy(), nll=fsup.detach().numpy(), loss_alone=float(loss),
                        dlogits=dlog.numpy(), lr=lr.numpy(), hr=hr.numpy(), loss=np.array(traj["loss"]), gnorm=np.array(traj["gnorm"]),
                        loss_ctc=np.array(traj["loss_ctc"]), prior_argmax_step0=prior_argmax)
    print("ctc_loss.npz written")


if __name__ == "__main__":
    main()
