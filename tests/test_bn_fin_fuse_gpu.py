"""GPU: BatchNorm finalize by the convolution launch that produces the statistics (tpgsr_conv_args.fin_mode): the whole-CU halo kernel's LAST
workgroup reduces the partial rows (write-through stores, one ticket per workgroup, L1-bypassing loads, fixed summation order) -- every
other kernel gets the reduction appended as a launch by tpgsr_conv_fwd.  Against the separate tpgsr_bn_finalize / tpgsr_bn_bwd_finalize
launches on the same partial rows: forward (scale, shift, saved mean / rstd, running statistics), backward (dgamma, dbeta accumulated,
the three coefficient rows), both on the kernel that fuses and on one that does not; repeated launches (the ticket counter returns to zero)
and bitwise run-to-run equality."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(N, H, W, Ci, Co, seed=0):
    from tpgsr_amd import kernels as K
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N * H * W, Ci, generator=g).to(DEV)
    w = (torch.randn(9 * Ci, Co, generator=g) / math.sqrt(9 * Ci)).to(DEV)
    geom = K.ConvGeom(N, H, W, Ci, Co, 3, 3, 1, 1)
    t = dict(gamma=(torch.rand(Co, generator=g) + 0.5).to(DEV), beta=torch.randn(Co, generator=g).to(DEV),
             bias=torch.randn(Co, generator=g).to(DEV), rm=torch.randn(Co, generator=g).to(DEV), rv=(torch.rand(Co, generator=g) + 0.5).to(DEV))
    return K, geom, x, w, t


@pytest.mark.parametrize("shape,halo3", [((48, 16, 64, 64, 64), True), ((48, 16, 64, 64, 64), False), ((48, 8, 25, 128, 256), True),
                                          ((13, 16, 64, 64, 64), True)])
def test_forward_finalize_by_the_convolution_launch(shape, halo3, monkeypatch):
    from tpgsr_amd import _lib, kernels
    monkeypatch.setattr(kernels, "BN_FIN_FUSE", True)         # (off by default: measured slower, profiles/r04aa_bn_fin_fuse_ab.md)
    lib = _lib.load()
    K, geom, x, w, t = _setup(*shape)
    Co, M = geom.Cout, geom.M
    nblk = (M + 63) // 64
    lib.tpgsr_halo3_set_enabled(1 if halo3 else 0)
    try:
        with K.conv_terms(2):
            K.make_bf_twin(w, geom.Cin)
            res = []
            for fused in (False, True, True):
                out = torch.empty(M, Co, device=DEV)
                part = torch.full((nblk, 2, Co), float("nan"), device=DEV)
                o = {k: torch.full((Co,), float("nan"), device=DEV) for k in ("scale", "shift", "mean", "rstd")}
                rm, rv = t["rm"].clone(), t["rv"].clone()
                counter = torch.zeros(1, dtype=torch.int32, device=DEV)
                if fused:
                    fin = dict(mode=1, count=M, counter=counter, gamma=t["gamma"], beta=t["beta"], bias=t["bias"], scale=o["scale"], shift=o["shift"],
                               save_mean=o["mean"], save_rstd=o["rstd"], running_mean=rm, running_var=rv)
                    args = K.make_conv_args(geom, x, w, out, bias=t["bias"], bn_partial=part, bn_fin=fin)
                    for _ in range(3):         # the counter must be back at zero after every launch
                        rm.copy_(t["rm"]); rv.copy_(t["rv"])
                        K.conv_fwd(args)
                else:
                    K.conv_fwd(K.make_conv_args(geom, x, w, out, bias=t["bias"], bn_partial=part))
                    K.bn_finalize(part, nblk, Co, M, t["bias"], t["gamma"], t["beta"], rm, rv, o["scale"], o["shift"], o["mean"], o["rstd"])
                torch.cuda.synchronize()
                assert int(counter.item()) == 0
                res.append(dict(out=out, rm=rm, rv=rv, **o))
    finally:
        lib.tpgsr_halo3_set_enabled(1)
    ref, a, b = res
    for k in ref:
        assert torch.equal(a[k], b[k]), k                      # run to run: bitwise
        tol = 0 if k == "out" else 2e-6
        d = (a[k] - ref[k]).abs().max().item()
        assert d <= tol * max(1.0, ref[k].abs().max().item()), (k, d)


@pytest.mark.parametrize("halo3", [True, False])
def test_backward_finalize_by_the_data_gradient_launch(halo3, monkeypatch):
    from tpgsr_amd import _lib, kernels
    monkeypatch.setattr(kernels, "BN_FIN_FUSE", True)
    lib = _lib.load()
    K, geom, x, w, t = _setup(48, 16, 64, 64, 64, seed=3)
    Co, M = geom.Cout, geom.M
    nblk = (M + 63) // 64
    g = torch.Generator().manual_seed(9)
    y = torch.randn(M, Co, generator=g).to(DEV)
    mean, rstd = (torch.randn(Co, generator=g) * 0.1).to(DEV), (torch.rand(Co, generator=g) + 0.5).to(DEV)
    bsc, bsh = (torch.rand(Co, generator=g) + 0.5).to(DEV), (torch.randn(Co, generator=g) * 0.2).to(DEV)
    lib.tpgsr_halo3_set_enabled(1 if halo3 else 0)
    try:
        with K.conv_terms(2):
            K.make_bf_twin(w, geom.Cin)
            res = []
            for fused in (False, True, True):
                out = torch.empty(M, Co, device=DEV)
                part = torch.full((nblk, 2, Co), float("nan"), device=DEV)
                coef = torch.full((3, Co), float("nan"), device=DEV)
                dgamma, dbeta = torch.ones(Co, device=DEV), torch.full((Co,), 2.0, device=DEV)
                counter = torch.zeros(1, dtype=torch.int32, device=DEV)
                bnb = dict(y=y, mean=mean, rstd=rstd, scale=bsc, shift=bsh, act="mish", partial=part)
                if fused:
                    bnb["fin"] = dict(mode=2, count=M, counter=counter, gamma=t["gamma"], coef=coef, dgamma=dgamma, dbeta=dbeta, accumulate=True)
                    K.conv_fwd(K.make_conv_args(geom, x, w, out, bnb=bnb))
                else:
                    K.conv_fwd(K.make_conv_args(geom, x, w, out, bnb=bnb))
                    K.bn_bwd_finalize(part, nblk, Co, M, t["gamma"], mean, rstd, dgamma, dbeta, coef, accumulate=True)
                torch.cuda.synchronize()
                assert int(counter.item()) == 0
                res.append(dict(out=out, coef=coef, dgamma=dgamma, dbeta=dbeta))
    finally:
        lib.tpgsr_halo3_set_enabled(1)
    ref, a, b = res
    for k in ref:
        assert torch.equal(a[k], b[k]), k
        d = (a[k] - ref[k]).abs().max().item()
        assert d <= (0 if k == "out" else 2e-6) * max(1.0, ref[k].abs().max().item()), (k, d)
