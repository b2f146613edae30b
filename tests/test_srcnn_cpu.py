"""CPU: BASELINE config C1 -- SRCNN bs 4, forward + MSE step (plumbing case, no GPU) vs the reference fixture."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import tpgsr_oracle as O


def test_srcnn_c1_step(golden_dir):
    from tpgsr_amd.model.srcnn import SRCNN
    g = np.load(os.path.join(golden_dir, "model_srcnn.npz"))
    t = np.load(os.path.join(golden_dir, "train_c1.npz"))
    net = SRCNN()
    net.load_state_dict(O.recipe_state_dict(O.srcnn_spec(), 104))
    lr, hr = torch.tensor(g["lr"]), torch.tensor(g["hr"])
    assert (net(lr) - torch.tensor(g["y"])).abs().max() < 2e-5
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.5, 0.999))
    for step in range(2):      # interfaces/super_resolution.py:409-424, srcnn branch: 3 channels, nn.MSELoss, clip, Adam
        loss = F.mse_loss(net(lr[:, :3]), hr[:, :3]).mean() * 100
        opt.zero_grad(); loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 0.25)
        opt.step()
        assert abs(loss.item() - t["loss"][step]) < 1e-5 * t["loss"][step]
