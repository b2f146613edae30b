"""GPU soak: the recurrent kernels of the SR network, 1000 launches each at full batch (bs 48: 768 / 3072 sequences, three scanning
waves per SIMD) NEXT TO a co-running load on a second stream -- every launch the same bits as the first.

Why (round 6, profiles/r06_gru_proj_root_cause.md): the staged one-launch GruBlock forward was not repeatable at
full batch -- one packed multiply-add of the scan (v_pk_fma_f32 ... op_sel:[0,1,0]: LOW half from the ODD register of a pair a broadcast
ds_read_b128 had just returned) read that register as zero in lanes 48-63, about once in 10^4 time steps when three scanning waves shared a
SIMD.  The scans take odd components through a v_mov now (gru_common.h: gru_dup_odd); this test is what would have caught the old form:
with the lab copy of the kernel (tools/lab/gp_staged_lab.hip, variant 0) 14 of 16 launches differed.
Reference: GruBlock, model/tsrn.py:491-508."""
import pytest
import torch

from test_gru_proj_gpu import _case

pytestmark = pytest.mark.gpu
DEV = "cuda"
N, H, W = 48, 16, 64
LAUNCHES = 1000


class _Load:
    """a second stream kept busy with streaming + matrix work while the kernel under test runs (what the weight-gradient stream does to
    the scans inside a train step)"""

    def __init__(self):
        from tpgsr_amd import kernels as K
        self.K = K
        self.side = torch.cuda.Stream()
        n = 1 << 24                                        # 64 MB per buffer: beyond the L2s
        self.a, self.b, self.c = (torch.randn(n, device=DEV) for _ in range(3))
        g = K.ConvGeom(N, H, W, 64, 64, 3, 3, 1, 1)
        self.x = torch.randn(g.M, 64, device=DEV)
        self.wf = torch.randn(g.K, 64, device=DEV) * 0.05
        self.out = torch.empty(g.M, 64, device=DEV)
        with K.conv_terms(2):
            K.make_bf_twin(self.wf, 64)
            self.conv = K.make_conv_args(g, self.x, self.wf, self.out)
        torch.cuda.synchronize()

    def kick(self):
        K = self.K
        with torch.cuda.stream(self.side), K.conv_terms(2):
            K.add(self.a, self.b, self.a.numel(), self.c)
            K.conv_fwd(self.conv)


def _soak(launch, outs_of, label):
    """launch(i) -> queues launch i writing into buffer set i % 2; outs_of(k) -> tuple of tensors of buffer set k"""
    load = _Load()
    launch(0)
    torch.cuda.synchronize()
    ref = tuple(t.clone() for t in outs_of(0))
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    for i in range(1, LAUNCHES + 1):
        if i % 4 == 1:
            load.kick()
        k = i % 2
        for t in outs_of(k):
            t.fill_(float("nan"))
        launch(i)
        for t, r in zip(outs_of(k), ref):
            bad += (t.view(torch.int32) != r.view(torch.int32)).any().to(torch.int64)
    torch.cuda.synchronize()
    print(f"{label}: {LAUNCHES} launches next to a co-running load, {int(bad)} differed from the first")
    assert int(bad) == 0, f"{label}: {int(bad)} of {LAUNCHES} launches differ bitwise from the first"


@pytest.mark.parametrize("terms", [2, 3])
@pytest.mark.parametrize("axis,loader", [(1, "affine+strip"), (1, "affine"), (0, "residual")])
def test_one_launch_gru_block_forward_soak(axis, loader, terms):
    from tpgsr_amd import kernels as K
    Cin = 96 if loader == "affine+strip" else 64
    t, kw = _case(N, H, W, Cin, loader, seed=5)
    P = N * H * W
    geom = K.ConvGeom(N, H, W, Cin, 192)
    bufs = [(torch.empty(P, 64, device=DEV), torch.empty(P, 256, device=DEV)) for _ in range(2)]
    with K.conv_terms(terms):
        K.make_bf_twin(t["wc"], 0)
        pas = [K.make_bigru_proj_args(K.make_conv_args(geom, t["x"], t["wc"], None, bias=t["bc"], **kw), t["whh"], t["bhh"], axis, h, gt) for h, gt in bufs]
        assert K.bigru_proj_supported(pas[0])

        def launch(i):
            with K.conv_terms(terms):
                K.bigru_proj_fwd(pas[i % 2])
        _soak(launch, lambda k: bufs[k], f"bigru_proj_fwd axis {axis} {loader} x{terms}")


@pytest.mark.parametrize("axis", [0, 1])
def test_scan_kernels_soak(axis):
    """the scan over a precomputed projection (tpgsr_bigru_fwd: the f32 policy's path, one wave per workgroup = three per SIMD along H) and
    back-propagation through time (tpgsr_bigru_bwd2)"""
    from tpgsr_amd import kernels as K
    g = torch.Generator().manual_seed(3 + axis)
    P = N * H * W
    gi = torch.randn(P, 192, generator=g).to(DEV)
    whh = (torch.randn(2, 96, 32, generator=g) / 32 ** 0.5).to(DEV)
    bhh = (torch.randn(2, 96, generator=g) * 0.1).to(DEV)
    dh = torch.randn(P, 64, generator=g).to(DEV)
    fb = [(torch.empty(P, 64, device=DEV), torch.empty(P, 256, device=DEV)) for _ in range(2)]
    _soak(lambda i: K.bigru_fwd(gi, whh, bhh, N, H, W, axis, *fb[i % 2]), lambda k: fb[k], f"bigru_fwd axis {axis}")
    h0, gt0 = fb[0][0].clone(), fb[0][1].clone()
    K.bigru_fwd(gi, whh, bhh, N, H, W, axis, h0, gt0)
    bb = [(torch.empty(P, 192, device=DEV), torch.empty(P, 64, device=DEV)) for _ in range(2)]
    _soak(lambda i: K.bigru_bwd2(gt0, h0, dh, None, whh, N, H, W, axis, *bb[i % 2]), lambda k: bb[k], f"bigru_bwd2 axis {axis}")
