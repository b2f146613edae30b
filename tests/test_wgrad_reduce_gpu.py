"""GPU: the slab reduce of the weight gradients (tpgsr_wgrad_reduce / tpgsr_wgrad_reduce_program, csrc/conv_mfma.hip) on random slabs,
independent of the kernels that produce them: sum over Z slabs [k = (tap, ci)][co] -> PyTorch [co][ci][kh][kw] (`layout` 0: the tiled,
transposing path; 9x9 kernels and Cout % 4 != 0: the linear path), padded operands (`real`), += accumulation with a gradient scale, bias
entries, several layers in one program launch.  Reference: fp64 sum of the same slabs; the kernel's own order of additions is fixed, so
two launches are also compared bitwise."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

SHAPES = [  # Cin, Cout, KH, KW, Z, cin_ld
    (64, 64, 3, 3, 37, 0),       # trunk conv: 3 input channels x 9 taps per tile, ragged last tile (64 = 21*3 + 1)
    (64, 256, 3, 3, 16, 0),      # up-sample conv
    (64, 192, 1, 1, 128, 0),     # GRU input weights
    (64, 192, 1, 3, 9, 0),       # 1x3
    (4, 64, 9, 9, 21, 0),        # head: linear path (81 taps)
    (3, 64, 3, 3, 5, 4),         # padded operand: cin_ld 4
    (64, 37, 1, 1, 7, 0),        # Cout % 4 != 0: linear path (recogniser's class layer)
    (512, 512, 3, 3, 4, 0),
    (32, 96, 1, 1, 128, 0),
    (130, 36, 3, 3, 3, 0),
]


def _ref(part, Z, Cin, Cout, KH, KW, cin_ld):
    ld = cin_ld or Cin
    s = part.double().view(Z, -1, Cout).sum(0)                 # [K][Cout]
    s = s[:KH * KW * ld].view(KH * KW, ld, Cout)[:, :Cin]       # [tap][ci][co]
    return s.permute(2, 1, 0).reshape(Cout, Cin, KH, KW)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("accumulate", [False, True])
def test_reduce_matches_fp64_sum(shape, accumulate):
    from tpgsr_amd import kernels as K
    Cin, Cout, KH, KW, Z, cin_ld = shape
    ld = cin_ld or Cin
    Kd = KH * KW * ld
    g = torch.Generator().manual_seed(Cin * 131 + Cout)
    part = torch.randn(Z * Kd * Cout, generator=g).to(DEV)
    dbp = torch.randn(Z * Cout, generator=g).to(DEV)
    dw0 = torch.randn(Cout, Cin, KH, KW, generator=g).to(DEV)
    db0 = torch.randn(Cout, generator=g).to(DEV)
    geom = K.ConvGeom(1, 8, 8, ld, Cout, KH, KW)
    assert geom.K == Kd
    gs = 0.37 if accumulate else 1.0
    outs = []
    for _ in range(2):
        dw, db = dw0.clone(), db0.clone()
        K.wgrad_reduce(part, dbp, Z, geom, dw, db, accumulate=accumulate, gscale=gs, real=(Cin, KH, KW, cin_ld) if cin_ld else None)
        torch.cuda.synchronize()
        outs.append((dw, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = _ref(part.cpu(), Z, Cin, Cout, KH, KW, cin_ld) * gs + (dw0.cpu().double() if accumulate else 0)
    refb = dbp.cpu().double().view(Z, Cout).sum(0) + (db0.cpu().double() if accumulate else 0)
    tol = 2e-6 * (Z ** 0.5) * 4 + 1e-6
    assert (outs[0][0].cpu().double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    assert (outs[0][1].cpu().double() - refb).abs().max().item() <= tol * max(1.0, refb.abs().max().item())


def test_program_of_many_layers_equals_single_launches():
    from tpgsr_amd import kernels as K
    items, singles = [], []
    g = torch.Generator().manual_seed(5)
    for Cin, Cout, KH, KW, Z, cin_ld in SHAPES:
        if cin_ld:
            continue
        Kd = KH * KW * Cin
        part = torch.randn(Z * Kd * Cout, generator=g).to(DEV)
        dbp = torch.randn(Z * Cout, generator=g).to(DEV)
        dw, db = torch.zeros(Cout, Cin, KH, KW, device=DEV), torch.zeros(Cout, device=DEV)
        dw1, db1 = torch.zeros_like(dw), torch.zeros_like(db)
        items.append((part, dbp, Z, Kd, Cin, Cout, KH, KW, 0, dw, db, 0, 1.0, 0))
        K.wgrad_reduce(part, dbp, Z, K.ConvGeom(1, 8, 8, Cin, Cout, KH, KW), dw1, db1, accumulate=False)
        singles.append((dw1, db1))
    K._reduce_program(items)
    torch.cuda.synchronize()
    for it, (dw1, db1) in zip(items, singles):
        assert torch.equal(it[9], dw1) and torch.equal(it[10], db1)
