"""GPU: tpgsr_conv_wgrad_batch (csrc/conv_xbf.hip) -- several independent 1x1 weight-gradient GEMMs in one launch, what the text-prior
generator's BiLSTM layers (model/crnn/crnn.py:5-26) hand the weight-gradient stream -- against the same GEMMs launched one by one: the
slabs must be bitwise the same (same workgroups, same order of additions), for the shapes of the two layers (strided operands: one
direction's hidden states / gate gradients inside the shared tensors, negative pad_w = the shifted previous state), a padded-row
embedding, an affine + ReLU loader batch, and mixed batches that split into two launches + a single one."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(K, g, N, T, Cin, Cout, *, in_ld=None, in_coff=0, dy_ld=None, dy_coff=0, shift=0, loader=False, terms=2):
    x = torch.randn(N * T, in_ld or Cin, generator=g).to(DEV)
    dy = torch.randn(N * T, dy_ld or Cout, generator=g).to(DEV)
    geom = K.ConvGeom(N, 1, T, Cin, Cout, 1, 1, 0, shift, 1, T)
    Z = K.wgrad_splits(geom.M, geom.K, Cout)
    kw = {}
    if loader:
        kw = dict(in_scale=(torch.rand(Cin, generator=g) + 0.5).to(DEV), in_shift=torch.randn(Cin, generator=g).to(DEV), in_act="relu")
    ca = K.make_conv_args(geom, x, in_ld=in_ld or Cin, in_coff=in_coff, **kw)

    def args():
        part = torch.full((Z, geom.K, Cout), float("nan"), device=DEV)
        dbp = torch.full((Z, Cout), float("nan"), device=DEV)
        return K.make_wgrad_args(ca, dy, part, dbp, dy_ld=dy_ld or Cout, dy_coff=dy_coff), part, dbp
    return args, (x, dy, kw)


@pytest.mark.parametrize("terms", [2, 3])
def test_batched_weight_gradients_equal_single_launches(terms):
    from tpgsr_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    N, T = 48, 26
    with K.conv_terms(terms):
        specs = [
            _mk(K, g, N, T, 512, 37, dy_ld=40),                                      # embedding, rows padded to 40
            _mk(K, g, N, T, 256, 1024, in_ld=512, in_coff=0, dy_ld=2048, dy_coff=0, shift=1),      # hidden side, forward direction
            _mk(K, g, N, T, 256, 1024, in_ld=512, in_coff=256, dy_ld=2048, dy_coff=1024, shift=-1),  # hidden side, reverse
            _mk(K, g, N, T, 256, 1024, dy_ld=2048, dy_coff=0),                       # input side of the second layer
            _mk(K, g, N, T, 512, 1024, dy_ld=2048, dy_coff=1024, loader=True),       # input side of the first layer: BatchNorm + ReLU loader
            _mk(K, g, N, T, 512, 1024, dy_ld=2048, dy_coff=0, loader=True),
            _mk(K, g, N, T, 512, 256),                                               # embedding of the first layer
        ]
        single, batched, keep = [], [], []
        for mk, k in specs:
            w, part, dbp = mk()
            K.conv_wgrad(w)
            single.append((part, dbp))
            keep.append((w, k))
        ws = []
        for mk, k in specs:
            w, part, dbp = mk()
            ws.append(w)
            batched.append((part, dbp))
        K.conv_wgrad_batch(ws)
        torch.cuda.synchronize()
    for i, ((p0, b0), (p1, b1)) in enumerate(zip(single, batched)):
        assert not torch.isnan(p0).any()
        assert torch.equal(p0, p1), (i, "slabs")
        assert torch.equal(b0, b1), (i, "bias slabs")
