"""CPU: `--tpg OPT` inside the training loop, oracle side: oracle/opt_oracle.py + tpgsr_train_step(tpg_forward=opt_forward) against the
fixture tests/golden/make_golden_next.py wrote from the reference's own modules (train_c3_opt.npz: two C3-shaped steps with
crnn.Model(opt) as teacher and student, interfaces/super_resolution.py:77-80, :295-424)."""
import os
import sys

import numpy as np
import torch

from oracle import opt_oracle as OO
from oracle import tpgsr_oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden_next import generic_recipe  # noqa: E402  (the weight recipe only; the reference is not imported here)

OPT = dict(Transformation="None", FeatureExtraction="ResNet", SequenceModeling="None", Prediction="CTC", num_fiducial=20,
           input_channel=1, output_channel=512, hidden_size=256, num_class=37)


def opt_state_dicts():
    from tpgsr_amd.model.crnn import model as opt
    tmpl = opt.Model(OPT).state_dict()
    return generic_recipe(tmpl, 312), generic_recipe(tmpl, 313)


def test_oracle_opt_train_step_vs_reference_fixture(golden_dir):
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    t = np.load(os.path.join(golden_dir, "train_c3_opt.npz"))
    lr, hr = torch.tensor(t["lr"]), torch.tensor(t["hr"])
    sd_sr = O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 301, tps_hw=(16, 64))
    sd_t, sd_s = opt_state_dicts()
    ps, pt, pu = O.as_params(sd_sr), O.as_params(sd_t, False), O.as_params(sd_s)
    opt = O.AdamState([ps[k] for k in O.trainable_keys(ps)] + [pu[k] for k in O.trainable_keys(pu)])
    for step in range(2):
        r = O.tpgsr_train_step([ps], [pu], pt, opt, lr, hr, stu_iter=1, tpg_forward=OO.opt_forward)
        tol = (1e-4, 5e-4)[step]
        assert abs(float(r["loss"]) - t["loss"][step]) <= tol * t["loss"][step]
        assert abs(float(r["grad_norms"][0]) - t["gnorm"][step]) <= 10 * tol * t["gnorm"][step]
        if step == 0:
            assert (r["priors"][0].argmax(-1).numpy() == t["prior_argmax_step0"]).all()
            assert (r["sr"] - torch.tensor(t["sr_step0"])).abs().max() < 2e-4
