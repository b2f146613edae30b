"""GPU: the BiGRU recurrence's own sigmoid / tanh (csrc/gru.hip: v_exp_f32 on a compensated argument, v_rcp_f32 + one Newton step, an odd
polynomial for tanh near 0) against fp64 over the whole range the gates can see -- they replaced libm's expf / expm1f / IEEE division in the
time-step loop, whose instruction count is its latency, and must stay at libm's accuracy (common.h records what a bare v_exp_f32 did to
the text-prior gradient: 6e-4 drift)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ulps(got, ref64):
    ref32 = ref64.float()
    # one ulp of the fp32 result at the reference's magnitude
    ulp = torch.maximum(torch.abs(ref32), torch.full_like(ref32, 2.0 ** -126)).double()
    ulp = 2.0 ** (torch.floor(torch.log2(ulp)) - 23)
    return ((got.double() - ref64).abs() / ulp).max().item()


def test_gate_math_matches_fp64():
    from tpgsr_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    xs = torch.cat([torch.randn(1 << 18, generator=g) * 3, torch.randn(1 << 16, generator=g) * 30, torch.randn(1 << 16, generator=g) * 1e-3,
                    torch.linspace(-0.3, 0.3, 1 << 16), torch.tensor([0.0, -0.0, 0.25, -0.25, 87.0, -87.0, 100.0, -100.0, 1e-30, -1e-30, 1e4, -1e4])])
    x = xs.cuda()
    sg, th = torch.empty_like(x), torch.empty_like(x)
    rc = lib.tpgsr_gru_gate_math_probe(C.c_void_p(x.data_ptr()), C.c_void_p(sg.data_ptr()), C.c_void_p(th.data_ptr()), x.numel(),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    xd = xs.double()
    sg_ref, th_ref = torch.sigmoid(xd), torch.tanh(xd)
    assert torch.isfinite(sg).all() and torch.isfinite(th).all()
    # relative accuracy down to sigmoid(-69) = 1e-30; below, the exponent's argument is clamped at 80 (1 / (1 + e^80) = 1.8e-35 is the
    # floor: the reciprocal's Newton step must not see an infinite denominator) -- an absolute test there
    big = sg_ref > 1e-30
    assert _ulps(sg.cpu()[big], sg_ref[big]) <= 4.0
    assert (sg.cpu()[~big].double() - sg_ref[~big]).abs().max().item() <= 2e-35
    assert _ulps(th.cpu(), th_ref) <= 5.0
    assert th.cpu()[xs == 0].abs().max().item() == 0.0
