"""GPU: tpgsr_gru_wgrad (csrc/gru_wgrad.hip) -- every weight gradient of a GruBlock (model/tsrn.py:491-508) in one launch -- against
fp64 contractions of the same operands and against the three tile-loop weight-gradient launches it replaces, for the three loader
shapes the SR network uses (residual add: gru2; BatchNorm affine: gru1; affine + concatenated text strip: gru1 of TSRN_TL), both scan
axes, ragged pixel counts, all three term counts."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _case(N, H, W, Cin, axis, loader, seed):
    g = torch.Generator().manual_seed(seed)
    P = N * H * W
    t = dict(dgi=torch.randn(P, 192, generator=g), dghn=torch.randn(P, 64, generator=g), h=torch.randn(P, 64, generator=g))
    if loader == "res":
        t["x"], t["x2"] = torch.randn(P, Cin, generator=g), torch.randn(P, Cin, generator=g)
        A = t["x"] + t["x2"]
    elif loader == "bn":
        t["x"] = torch.randn(P, Cin, generator=g)
        t["sc"], t["sh"] = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
        A = t["x"] * t["sc"] + t["sh"]
    elif loader == "bn_strip":
        t["x"] = torch.randn(P, 64, generator=g)
        t["strip"] = torch.randn(N * W, Cin - 64, generator=g)
        t["sc"], t["sh"] = torch.ones(Cin), torch.zeros(Cin)
        t["sc"][:64], t["sh"][:64] = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
        img = t["x"] * t["sc"][:64] + t["sh"][:64]
        strip = t["strip"].view(N, 1, W, Cin - 64).expand(N, H, W, Cin - 64).reshape(P, Cin - 64)
        A = torch.cat([img, strip], 1)
    else:
        t["x"] = torch.randn(P, Cin, generator=g)
        A = t["x"]
    # fp64 reference
    A, dgi, dghn, h = A.double(), t["dgi"].double(), t["dghn"].double(), t["h"].double()
    h4 = h.view(N, H, W, 64)
    ref = dict(dWc=dgi.t() @ A, dbc=dgi.sum(0))                      # [192][Cin]
    for d in range(2):
        hp = torch.zeros_like(h4)
        sl = slice(d * 32, d * 32 + 32)
        if axis == 0:
            if d == 0:
                hp[:, :, 1:, sl] = h4[:, :, :-1, sl]
            else:
                hp[:, :, :-1, sl] = h4[:, :, 1:, sl]
        else:
            if d == 0:
                hp[:, 1:, :, sl] = h4[:, :-1, :, sl]
            else:
                hp[:, :-1, :, sl] = h4[:, 1:, :, sl]
        hp = hp.reshape(P, 64)[:, sl]
        dgh = torch.cat([dgi[:, d * 96:d * 96 + 64], dghn[:, d * 32:d * 32 + 32]], 1)      # (dr, dz, dn * r)
        ref[f"dWhh{d}"] = dgh.t() @ hp                                 # [96][32] = weight_hh_l0's layout
        ref[f"dbhh{d}"] = dgh.sum(0)
    return t, ref


def _run(t, N, H, W, Cin, axis, loader, terms):
    from tpgsr_amd import kernels as K
    from tpgsr_amd.kernels import ConvGeom
    d = {k: v.to(DEV).contiguous() for k, v in t.items()}
    P = N * H * W
    kw = {}
    if loader == "res":
        kw = dict(in2=d["x2"])
    elif loader == "bn":
        kw = dict(in_scale=d["sc"], in_shift=d["sh"])
    elif loader == "bn_strip":
        kw = dict(in_scale=d["sc"], in_shift=d["sh"], in_b=d["strip"], cin_a=64)
    with K.conv_terms(terms):
        Z = K.gru_wgrad_splits(P)
        partC, dbC = torch.full((Z * Cin * 192,), float("nan"), device=DEV), torch.full((Z * 192,), float("nan"), device=DEV)
        partH, dbH = torch.full((2 * Z * 32 * 96,), float("nan"), device=DEV), torch.full((2 * Z * 96,), float("nan"), device=DEV)
        gc = ConvGeom(N, H, W, Cin, 192)
        K.gru_wgrad(K.make_conv_args(gc, d["x"], **kw), d["dgi"], d["dghn"], d["h"], axis, Z, partC, dbC, partH, dbH)
        out = dict(dWc=torch.zeros(192, Cin, device=DEV), dbc=torch.zeros(192, device=DEV))
        K.wgrad_reduce(partC, dbC, Z, gc, out["dWc"], out["dbc"], accumulate=False)
        for dd in range(2):
            out[f"dWhh{dd}"], out[f"dbhh{dd}"] = torch.zeros(96, 32, device=DEV), torch.zeros(96, device=DEV)
            K.wgrad_reduce(partH[dd * Z * 3072:(dd + 1) * Z * 3072], dbH[dd * Z * 96:(dd + 1) * Z * 96], Z, ConvGeom(N, H, W, 32, 96),
                           out[f"dWhh{dd}"], out[f"dbhh{dd}"], accumulate=False)
    torch.cuda.synchronize()
    return {k: v.double().cpu() for k, v in out.items()}


@pytest.mark.parametrize("N,H,W,Cin,axis,loader", [
    (48, 16, 64, 64, 0, "res"),          # gru2 of every block at the bench batch size
    (48, 16, 64, 96, 1, "bn_strip"),     # gru1 of TSRN_TL
    (48, 16, 64, 64, 1, "bn"),           # gru1 of TSRN
    (3, 16, 64, 96, 0, "bn_strip"),
    (2, 5, 7, 64, 1, "plain"),           # ragged: 70 pixels, sequences of 5
    (1, 3, 11, 64, 0, "res"),            # 33 pixels: one full chunk + one pixel
])
@pytest.mark.parametrize("terms", [3, 2, 1])
def test_gru_wgrad_vs_fp64(N, H, W, Cin, axis, loader, terms):
    t, ref = _case(N, H, W, Cin, axis, loader, seed=N * 1000 + Cin + axis)
    got = _run(t, N, H, W, Cin, axis, loader, terms)
    tol = {3: 2e-6, 2: 3e-5, 1: 6e-3}[terms]
    for k, r in ref.items():
        err = ((got[k] - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item()
        assert err < tol, (k, err)
    if terms == 3:                        # bias gradients are fp32 column sums: tight in every mode
        for k in ("dbc", "dbhh0", "dbhh1"):
            assert (got[k] - ref[k]).abs().max() < 1e-5 * ref[k].abs().max() + 1e-4


def test_gru_block_backward_fused_equals_three_launches_to_rounding():
    """the SR network's GruLayer.bwd with the fused launch against the three tile-loop launches (TPGSR_GRU_WGRAD=0 path) on one TSRN_TL
    backward pass: every gradient of every GruBlock agrees to accumulation-order rounding"""
    from oracle import tpgsr_oracle as O
    from tpgsr_amd import kernels as K
    from tpgsr_amd.loss.image_loss import ImageLoss
    from tpgsr_amd.model import tsrn
    sd = O.recipe_state_dict(O.tsrn_spec(STN=False, mask=True, text_prior=True), 9, tps_hw=(16, 64))
    lr, hr = O.synthetic_batch(3, 6)
    prior = torch.softmax(torch.randn(3, 37, 1, 26, generator=torch.Generator().manual_seed(4)) * 2, 1)
    grads = []
    prev = K.GRU_WGRAD
    try:
        for fused in (True, False):
            K.GRU_WGRAD = fused
            net = tsrn.TSRN_TL(STN=False, mask=True)
            net.load_state_dict(sd)
            net = net.to(DEV).train()
            sr = net(lr.to(DEV), prior.to(DEV))
            (ImageLoss(gradient=True, loss_weight=[1, 1e-4])(sr, hr.to(DEV)).mean() * 100).backward()
            torch.cuda.synchronize()
            grads.append({n: p.grad.detach().double().cpu().clone() for n, p in net.named_parameters()})
            names = [op[0] for pl in net._engine()._plans.values() for op in pl["bwd"].ops]
            assert ("tpgsr_gru_wgrad" in names) == fused
    finally:
        K.GRU_WGRAD = prev
    worst = 0.0
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        rel = (a - b).norm().item() / max(b.norm().item(), 1e-12)
        worst = max(worst, rel)
        assert rel < 2e-5, (n, rel)
    print(f"fused vs three launches: worst relative difference over all parameters {worst:.2e}")
