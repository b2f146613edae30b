"""GPU: the three-stream schedule of the TPGSR train step against its own SERIALISED execution, and under schedule fuzzing.

`TPGSRTrainStep` runs one fixed dependency graph over three HIP streams (caller's stream; weight-gradient stream; teacher / leaf stream),
with deferred joins, side batches, a forward prologue on the weight-gradient stream and gradient buckets launched from it.  Run-to-run
equality cannot see a missing edge that resolves the same way every time, so the reference here is the SAME step replayed in recording
order on ONE stream (`kernels.set_schedule(serial=True)`: tpgsr_plan_run3 drops every stream edge and sends every launch to the caller's
stream, and the step's own side / teacher sections run inline) -- which has no concurrency to get wrong.  Then the default schedule
must also survive spin kernels of random length around every fork / join / edge / wait_stream (10 seeds), and a co-running busy kernel
that changes the wave timing inside every kernel of the step (the condition under which round 3's deleted `teacher_late` test once
differed: DESIGN section 5).  Replaces nothing in the reference (it has one stream, interfaces/super_resolution.py:297-424)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tpgsr_oracle as O  # noqa: E402

DEV = "cuda"
STEPS = 4


def _reset(ts, sds):
    """back to step 0: parameters + BatchNorm buffers from the recipe state dicts, Adam moments and step counters zeroed"""
    mods = ts.sr + ts.stu
    for m, sd in zip(mods, sds):
        m.load_state_dict(sd)
    for st in ts.opt.state.values():
        st["m"].zero_()
        st["v"].zero_()
        st["step"].zero_()
    torch.cuda.synchronize()


def _run(ts, lr, hr, steps):
    losses = [ts.step(lr, hr) for _ in range(steps)]
    torch.cuda.synchronize()
    return [l.item() for l in losses], ts.pool.flat.clone(), ts.pool.grad.clone()


def _build(n_stu, stu_iter, bs, seed):
    import test_fullsize_gpu as T
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    sr, stus, teacher, sd_sr, sd_s, _ = T._tpgsr(n_stu, seeds=(31, 32, 33))
    ts = TPGSRTrainStep([sr], stus, teacher, stu_iter=stu_iter, sr_share=True, tpg_share=False)
    lr, hr = O.synthetic_batch(bs, seed)
    return ts, [sd_sr] + sd_s, lr.to(DEV), hr.to(DEV)


@pytest.fixture()
def schedule():
    from tpgsr_amd import kernels as K
    yield K
    K.set_schedule()          # always back to the default schedule
    torch.cuda.synchronize()


def test_c3_bs48_three_stream_step_bitwise_equals_serial_schedule(schedule):
    """C3 at bs 48, 4 steps: default (three streams) == serial (one stream, recording order), losses, parameters and gradients bit for bit.
    The serial replica records its OWN plans in serial mode (fresh networks), so recording-time stream decisions are covered too."""
    K = schedule
    ts, sds, lr, hr = _build(1, 1, 48, 1234)
    la, pa, ga = _run(ts, lr, hr, STEPS)
    K.set_schedule(serial=True)
    ts2, _, _, _ = _build(1, 1, 48, 1234)
    lb, pb, gb = _run(ts2, lr, hr, STEPS)
    K.set_schedule()
    # ... and the three-stream plans themselves replayed serially (same recorded plans, other executor mode)
    _reset(ts, sds)
    K.set_schedule(serial=True)
    lc, pc, gc = _run(ts, lr, hr, STEPS)
    print("default", la, "serial (own plans)", lb, "serial (same plans)", lc)
    assert la == lb == lc
    assert torch.equal(pa, pb) and torch.equal(ga, gb)
    assert torch.equal(pa, pc) and torch.equal(ga, gc)


def test_c5_shape_cascade_bitwise_equals_serial_schedule(schedule):
    """stu_iter 3, sr_share, three students, bs 32 (the C5 rank workload): three SR forwards / backwards per step through slots, the
    cascade's gradient path through parse_crnn_data, three prologues on the weight-gradient stream -- 2 steps, default == serial"""
    K = schedule
    ts, sds, lr, hr = _build(3, 3, 32, 555)
    la, pa, ga = _run(ts, lr, hr, 2)
    _reset(ts, sds)
    K.set_schedule(serial=True)
    lb, pb, gb = _run(ts, lr, hr, 2)
    print("default", la, "serial", lb)
    assert la == lb
    assert torch.equal(pa, pb) and torch.equal(ga, gb)


@pytest.mark.parametrize("noise", [0, 384])
def test_c3_bs48_schedule_fuzz_bitwise(schedule, noise):
    """10 seeds: a spin kernel of 0..60 us delays a random stream, the source and the destination of EVERY stream edge (plan forks /
    joins / edges and the step's own wait_stream calls); noise = 384: additionally a 384-workgroup busy kernel co-runs on a fourth
    stream.  Every seed must end in bitwise the serial schedule's losses, parameters and gradients."""
    K = schedule
    ts, sds, lr, hr = _build(1, 1, 48, 77)
    _run(ts, lr, hr, 1)                      # the plans are recorded under the default schedule
    _reset(ts, sds)
    K.set_schedule(serial=True)
    ref = _run(ts, lr, hr, 3)
    bad = []
    for seed in range(1, 11):
        _reset(ts, sds)
        K.set_schedule(fuzz_us=60, seed=seed, noise_blocks=noise)
        got = _run(ts, lr, hr, 3)
        K.set_schedule()
        if got[0] != ref[0] or not torch.equal(got[1], ref[1]) or not torch.equal(got[2], ref[2]):
            bad.append((seed, got[0], ref[0], int((got[2] != ref[2]).sum())))
    assert not bad, bad


def test_c2_step_bitwise_equals_serial_schedule(schedule):
    """TSRNTrainStep (C2, bs 48): main + weight-gradient + leaf streams inside one backward plan"""
    from tpgsr_amd.interfaces.super_resolution import TSRNTrainStep
    from tpgsr_amd.model import tsrn
    K = schedule
    sd = O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True), 1234, tps_hw=(16, 64))
    lr, hr = O.synthetic_batch(48, 9)
    lr, hr = lr.to(DEV), hr.to(DEV)
    out = []
    for serial in (False, True):
        net = tsrn.TSRN(STN=True, mask=True)
        net.load_state_dict(sd)
        K.set_schedule(serial=serial)
        ts = TSRNTrainStep(net.to(DEV).train())
        ls = [ts.step(lr, hr) for _ in range(3)]
        torch.cuda.synchronize()
        out.append(([l.item() for l in ls], ts.pool.flat.clone()))
        K.set_schedule()
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])


def test_tail_bwd_bias_partials_do_not_depend_on_wave_timing():
    """tpgsr_tail_bwd's per-block bias-gradient partials (model/tsrn.py:159, the 9x9 tail's bias): summed in a fixed order.  Until
    round 4 they were accumulated with an LDS float atomicAdd per thread, i.e. in wave-arrival order -- the one place of the step whose
    bits could follow the timing.  40 launches next to a co-running busy kernel give bitwise one result."""
    from tpgsr_amd import _lib, kernels as K
    lib = _lib.load()
    N, H, W, Co, KS = 48, 32, 128, 4, 9
    g = torch.Generator().manual_seed(3)
    out = torch.tanh(torch.randn(N, Co, H, W, generator=g)).to(DEV)
    dout = torch.randn(N, Co, H, W, generator=g).to(DEV)
    nblk = K.tail_bwd_blocks(N, H, W, Co, KS)
    dP = torch.empty(N * H * W, KS * Co, device=DEV)
    ref = None
    noise = torch.cuda.Stream()
    for it in range(40):
        dbp = torch.zeros(nblk * Co, device=DEV)
        if it:
            _lib.check(lib.tpgsr_spin(128 + 64 * (it % 5), 256, 200.0, 1, noise.cuda_stream), "tpgsr_spin")
        K.tail_bwd(out, dout, N, H, W, Co, KS, dP, dbp, nblk)
        torch.cuda.synchronize()
        if ref is None:
            ref = dbp.clone()
            want = (dout * (1 - out * out)).double().sum((0, 2, 3)).cpu()
            got = dbp.view(nblk, Co).double().sum(0).cpu()
            assert (got - want).abs().max() < 1e-6 * want.abs().max() + 1e-3
        assert torch.equal(dbp, ref), f"launch {it}: {(dbp != ref).sum().item()} partials differ"
