"""GPU: CRNN text-prior generator (conv stack + 2 BiLSTMs) and its satellite kernels against the reference fixtures /
the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import tpgsr_oracle as O  # noqa: E402

DEV = "cuda"


def _build(seed=103):
    from tpgsr_amd.model.crnn import crnn
    sd = O.recipe_state_dict(O.crnn_spec(), seed)
    net = crnn.CRNN(32, 1, 37, 256)
    net.load_state_dict(sd, strict=True)
    return net.to(DEV), sd


def test_parse_crnn_data_kernel(golden_dir):
    from tpgsr_amd import kernels as K
    g = np.load(os.path.join(golden_dir, "model_crnn.npz"))
    hr = torch.tensor(g["hr"]).to(DEV)
    N, C, H, W = hr.shape
    out = torch.empty(N, 1, 32, 100, device=DEV)
    K.bicubic_gray_fwd(hr, N, C, H, W, 32, 100, out)
    torch.cuda.synchronize()
    assert (out.cpu() - torch.tensor(g["gray"])).abs().max() < 2e-6
    # adjoint vs autograd
    x = torch.rand(2, 4, 16, 64)
    xr = x.clone().requires_grad_(True)
    y = O.parse_crnn_data(xr)
    gy = torch.randn(y.shape)
    y.backward(gy)
    din = torch.empty(2, 4, 16, 64, device=DEV)
    gyd = gy.to(DEV).contiguous()
    K.bicubic_gray_bwd(gyd, 2, 4, 16, 64, 32, 100, din)
    torch.cuda.synchronize()
    assert (din.cpu() - xr.grad).abs().max() < 1e-5


@pytest.mark.parametrize("cfg", [((2, 2), (2, 2), (0, 0)), ((2, 2), (2, 1), (0, 1))])
def test_pool2d(cfg):
    from tpgsr_amd import kernels as K
    k, s, pd = cfg
    N, H, W, C = 2, 4, 9, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, C, H, W, generator=g)
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.3
    z = (x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).double().requires_grad_(True)
    o = F.max_pool2d(F.relu(z), k, s, pd)
    do = torch.randn(o.shape, generator=g)
    o.backward(do.double())
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    scd, shd = sc.to(DEV), sh.to(DEV)
    OH, OW = o.shape[2], o.shape[3]
    out = torch.empty(N * OH * OW, C, device=DEV)
    K.pool2d_fwd(xd, N, H, W, C, scd, shd, "relu", k, s, pd, out)
    dod = do.permute(0, 2, 3, 1).contiguous().to(DEV)
    dz = torch.empty(N * H * W, C, device=DEV)
    K.pool2d_bwd(xd, dod, N, H, W, C, scd, shd, "relu", k, s, pd, dz)
    torch.cuda.synchronize()
    assert (out.reshape(N, OH, OW, C).permute(0, 3, 1, 2).cpu() - o.detach()).abs().max() < 1e-6
    assert (dz.reshape(N, H, W, C).permute(0, 3, 1, 2).cpu() - z.grad).abs().max() < 1e-6


def test_softmax_prior_semantic_loss(golden_dir):
    from tpgsr_amd import kernels as K
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    p_ref, q = torch.tensor(g["p"]), torch.tensor(g["q"])           # (T, N, C) probabilities
    T, N, C = p_ref.shape
    logits = torch.log(p_ref) + 0.37                                 # any logits with softmax == p_ref
    lg = logits.permute(1, 0, 2).contiguous().to(DEV)                # [N][T][C]
    qd = q.permute(1, 0, 2).contiguous().to(DEV)
    p = torch.empty(N, T, C, device=DEV); prior = torch.empty(N, C, 1, T, device=DEV)
    nblk = 8
    part = torch.empty(nblk, 2, device=DEV); loss = torch.empty((), device=DEV)
    K.softmax_prior_fwd(lg, qd, N, T, C, 1, p, prior, part, nblk)
    K.semantic_loss_finalize(part, nblk, N * T * C, 1.0, loss)
    torch.cuda.synchronize()
    assert (p.cpu().permute(1, 0, 2) - p_ref).abs().max() < 1e-6
    assert abs(loss.item() - float(g["semantic_loss"])) < 1e-5 * float(g["semantic_loss"])
    ref_prior = p_ref.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2).clone()
    ref_prior[:1] = 0
    assert (prior.cpu() - ref_prior).abs().max() < 1e-6
    # backward: semantic-loss gradient (weight 1) + an arbitrary prior gradient, through the softmax
    lr_ = logits.clone().requires_grad_(True)
    pv = torch.softmax(lr_, -1)
    dprior = torch.randn(N, C, 1, T)
    pf = pv.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
    mask = torch.ones(N); mask[:1] = 0
    (O.semantic_loss(pv, q) + (pf * mask.view(-1, 1, 1, 1) * dprior).sum()).backward()
    dl = torch.empty(N, T, C, device=DEV)
    dpd = dprior.to(DEV).contiguous()
    K.softmax_prior_bwd(p, qd, dpd, None, N, T, C, 1, 1.0, dl, nblk)
    torch.cuda.synchronize()
    assert (dl.cpu().permute(1, 0, 2) - lr_.grad).abs().max() < 2e-6 * max(1.0, lr_.grad.abs().max().item())


def test_crnn_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "model_crnn.npz"))
    net, sd = _build()
    gray = torch.tensor(g["gray"]).to(DEV)
    gl = torch.tensor(g["gl"]).to(DEV)
    net.train()
    y = net(gray)
    assert tuple(y.shape) == (26, 2, 37)
    err = (y.detach().cpu() - torch.tensor(g["y_train"])).abs().max().item()
    print("crnn train fwd max err", err)
    assert err < 1e-4
    (y * gl).sum().backward()
    P = dict(net.named_parameters())
    gmax = g["grad_norms"].max()
    worst = 0
    for n, ref_norm, head in zip([str(n) for n in g["grad_names"]], g["grad_norms"], g["grad_heads"]):
        got = P[n].grad.detach().cpu()
        e = abs(got.double().norm().item() - ref_norm) / max(ref_norm, 1e-3 * gmax)
        worst = max(worst, e)
        assert e < 5e-3, (n, e, ref_norm)
        k = min(8, got.numel())
        assert (got.reshape(-1)[:k] - torch.tensor(head[:k])).abs().max() < 5e-3 * max(ref_norm / np.sqrt(got.numel()), 1e-3 * gmax / np.sqrt(got.numel())) * 10, n
    print("crnn worst grad-norm rel err", worst)
    net2, _ = _build()
    net2.eval()
    with torch.no_grad():
        ye = net2(gray)
    err = (ye.cpu() - torch.tensor(g["y_eval"])).abs().max().item()
    print("crnn eval fwd max err", err)
    assert err < 1e-4
    assert (ye.cpu().argmax(-1) == torch.tensor(g["y_eval"]).argmax(-1)).all()      # identical arg-max text prior


def test_crnn_gradients_vs_oracle():
    net, sd = _build(seed=19)
    lr, hr = O.synthetic_batch(3, 8)
    gray = O.parse_crnn_data(lr)
    p = O.as_params(sd)
    gr = gray.clone().requires_grad_(True)
    y = O.crnn_forward(p, gr, training=True)
    gl = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    (y * gl).sum().backward()
    net.train()
    gd = gray.to(DEV).requires_grad_(True)
    yd = net(gd)
    (yd * gl.to(DEV)).sum().backward()
    assert (yd.detach().cpu() - y.detach()).abs().max() < 1e-4
    gmax = max(v.grad.norm().item() for v in p.values() if v.grad is not None)
    bad = []
    for n, q in net.named_parameters():
        ref = p[n].grad
        rel = (q.grad.cpu() - ref).norm().item() / max(ref.norm().item(), 1e-3 * gmax)
        if rel > 3e-3:
            bad.append((n, rel))
    assert not bad, bad[:10]
    rel = (gd.grad.cpu() - gr.grad).norm().item() / gr.grad.norm().item()
    print("dgray rel err", rel)
    assert rel < 3e-3
