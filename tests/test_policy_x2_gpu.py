"""GPU: the two-term-split arithmetic policies (tpgsr_amd/kernels.py) against the fp32 ORACLE on the gates `north_star` states --
|dPSNR| < 1e-3 dB and IDENTICAL arg-max text priors -- at the full C3 batch size on three batches, plus what a two-term weight
gradient costs in accuracy next to an fp32 GEMM's own accumulation noise (both against fp64).
  x3b2: forward x3 (fp32-equivalent), backward GEMMs two-term;  x2: SR net two-term as well, text-prior generator forward x3."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tpgsr_oracle as O  # noqa: E402

DEV = "cuda"


def _policy(name):
    from tpgsr_amd import kernels as K
    prev = K.POLICY
    K.set_conv_prec(name)
    return prev


@pytest.mark.parametrize("policy", ["x3b2", "x2"])
@pytest.mark.parametrize("seed", [1234, 77, 4242])
def test_c3_bs48_two_term_policies_hold_the_north_star_gates(policy, seed):
    import test_fullsize_gpu as T
    from tpgsr_amd import kernels as K
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    prev = _policy(policy)
    try:
        T._threads()
        sr, stus, teacher, sd_sr, sd_s, sd_t = T._tpgsr(1)
        lr, hr = O.synthetic_batch(48, seed)
        ts = TPGSRTrainStep([sr], stus, teacher, stu_iter=1)
        loss = ts.step(lr.to(DEV), hr.to(DEV))
        torch.cuda.synchronize()
    finally:
        K.set_conv_prec(prev)
    ps, pt, pu = O.as_params(sd_sr), O.as_params(sd_t, False), [O.as_params(x) for x in sd_s]
    opt = O.AdamState([ps[k] for k in O.trainable_keys(ps)] + [q[k] for q in pu for k in O.trainable_keys(q)])
    ref = O.tpgsr_train_step([ps], pu, pt, opt, lr, hr, stu_iter=1)
    dpsnr = abs(T._psnr(ts.last_sr, hr) - T._psnr(ref["sr"], hr))
    am = ts.last_p.cpu().permute(1, 0, 2).argmax(-1)
    mism = int((am != ref["priors"][0].argmax(-1)).sum())
    gn, gn_ref = ts.opt.grad_norm(sr).item(), float(ref["grad_norms"][0])
    print(f"C3 bs48 {policy} policy (seed {seed}): loss {loss.item():.6f} vs {ref['loss'].item():.6f}; |dPSNR| {dpsnr:.3e} dB; "
          f"arg-max mismatches {mism} / {am.numel()}; SR grad norm {gn:.4f} vs {gn_ref:.4f}")
    assert dpsnr < 1e-3                      # the north_star gate itself
    assert mism == 0
    assert abs(loss.item() - ref["loss"].item()) < 2e-5 * ref["loss"].item()
    assert abs(gn - gn_ref) < 2e-3 * gn_ref


def test_two_term_weight_gradient_error_next_to_fp32_accumulation_noise():
    """trunk weight gradient (M = 49152 pixels, 3x3, 64 -> 64) against fp64: fp32 matrix cores, x3 and the two-term split"""
    from tpgsr_amd import kernels as K
    from tpgsr_amd.kernels import ConvGeom
    g = torch.Generator().manual_seed(5)
    N, H, W, C = 48, 16, 64, 64
    x = torch.randn(N, H, W, C, generator=g)
    dy = torch.randn(N, H, W, C, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (C, C, 3, 3), dy.permute(0, 3, 1, 2).double(), padding=1)
    xd, dyd = x.to(DEV).reshape(-1, C).contiguous(), dy.to(DEV).reshape(-1, C).contiguous()
    geom = ConvGeom(N, H, W, C, C, 3, 3, 1, 1)
    errs = {}
    for name, terms in (("f32", 0), ("x3", 3), ("x2", 2)):
        with K.conv_terms(terms):
            Z = K.wgrad_splits(geom.M, geom.K, C)
            part = torch.empty(Z * geom.K * C, device=DEV)
            dw = torch.zeros(C, C, 3, 3, device=DEV)
            K.conv_wgrad(K.make_wgrad_args(K.make_conv_args(geom, xd), dyd, part, None))
            K.wgrad_reduce(part, None, Z, geom, dw, None, accumulate=False)
        torch.cuda.synchronize()
        d = dw.double().cpu() - ref
        errs[name] = (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print("weight-gradient rms error vs fp64:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["x3"] < 2e-6 and errs["f32"] < 2e-6
    assert errs["x2"] < 2e-5           # ~16 significand bits per operand
