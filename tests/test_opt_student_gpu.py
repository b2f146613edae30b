"""GPU: `--tpg OPT` as a first-class text-prior generator of the fused train step: TPGSRTrainStep([TSRN_TL], [Model(opt)],
Model(opt)) -- the reference selects crnn.Model(opt) for teacher and students of the same loop (interfaces/super_resolution.py:77-80,
interfaces/base.py:681-756) -- against the fixture written from the reference's own modules (train_c3_opt.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import opt_oracle as OO  # noqa: E402
from oracle import tpgsr_oracle as O  # noqa: E402
from test_opt_student_cpu import OPT, opt_state_dicts  # noqa: E402

DEV = "cuda"


def _build():
    from tpgsr_amd.model import tsrn
    from tpgsr_amd.model.crnn import model as opt
    sd_sr = O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 301, tps_hw=(16, 64))
    sd_t, sd_s = opt_state_dicts()
    sr = tsrn.TSRN_TL(STN=True, mask=True)
    sr.load_state_dict(sd_sr)
    teacher, stu = opt.Model(OPT), opt.Model(OPT)
    teacher.load_state_dict(sd_t)
    stu.load_state_dict(sd_s)
    return sr.to(DEV).train(), stu.to(DEV).train(), teacher.to(DEV).eval(), (sd_sr, sd_t, sd_s)


def test_opt_student_train_step_vs_reference_fixture(golden_dir):
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    t = np.load(os.path.join(golden_dir, "train_c3_opt.npz"))
    lr, hr = torch.tensor(t["lr"]).to(DEV), torch.tensor(t["hr"]).to(DEV)
    sr, stu, teacher, _ = _build()
    ts = TPGSRTrainStep([sr], [stu], teacher, stu_iter=1)
    losses, gns = [], []
    for step in range(2):
        loss = ts.step(lr, hr)
        torch.cuda.synchronize()
        losses.append(loss.item())
        gns.append(ts.opt.grad_norm(sr).item())
        if step == 0:
            assert (ts.last_p.cpu().permute(1, 0, 2).argmax(-1).numpy() == t["prior_argmax_step0"]).all()
            assert (ts.last_sr.cpu() - torch.tensor(t["sr_step0"])).abs().max() < 5e-3        # STN conditioning, as for the CRNN prior
    print("C3 / OPT losses", losses, t["loss"], "SR grad norms", gns, t["gnorm"])
    assert abs(losses[0] - t["loss"][0]) < 3e-4 * t["loss"][0]
    assert abs(gns[0] - t["gnorm"][0]) < 3e-3 * t["gnorm"][0]
    assert abs(losses[1] - t["loss"][1]) < 2e-2 * t["loss"][1]


def test_opt_student_gradients_vs_oracle():
    """raw gradients of one step (no optimiser) against the oracle, SR net and the OPT student, incl. what reaches the student
    through the text prior (it is not detached: interfaces/super_resolution.py:318-385)"""
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    sr, stu, teacher, (sd_sr, sd_t, sd_s) = _build()
    lr, hr = O.synthetic_batch(4, 99)
    ts = TPGSRTrainStep([sr], [stu], teacher, stu_iter=1)
    ts.pool.bind(torch.device(DEV, 0))
    teacher._engine().bind(torch.device(DEV, 0))
    loss = ts._phase_a(lr.to(DEV), hr.to(DEV))
    torch.cuda.synchronize()
    ps, pt, pu = O.as_params(sd_sr), O.as_params(sd_t, False), O.as_params(sd_s)
    opt = O.AdamState([ps[k] for k in O.trainable_keys(ps)] + [pu[k] for k in O.trainable_keys(pu)])
    ref = O.tpgsr_train_step([ps], [pu], pt, opt, lr, hr, stu_iter=1, tpg_forward=OO.opt_forward)
    assert abs(loss.item() - ref["loss"].item()) < 3e-4 * ref["loss"].item()
    keys = O.trainable_keys(pu)
    n_sr = len(O.trainable_keys(ps))
    Ps = dict(stu.named_parameters())
    num = den = 0.0
    for k, gref in zip(keys, ref["grads"][n_sr:]):
        d = Ps[k].grad.cpu() - gref
        num += d.double().pow(2).sum().item()
        den += gref.double().pow(2).sum().item()
    print("OPT student gradients: global rel err", (num / den) ** 0.5)
    assert (num / den) ** 0.5 < 2e-2          # below the max-pools one flipped window moves everything under it
    # the student's parameters live in the pooled arena like the SR network's
    a, b = ts.pool.ranges[id(stu)]
    assert ts.pool.flat[a:b].data_ptr() == stu._engine().arena.flat.data_ptr()
