"""GPU: the BASELINE.json configurations at their FULL sizes against the CPU oracle itself (not properties):
C2 (TSRN, bs 48), C3 (TPGSR-TSRN + CRNN prior, bs 48, = one rank of C4) and a C5-shaped step (stu_iter 3, sr_share,
three students, bs 32).  Gates as `north_star` states them: |dPSNR| < 1e-3 dB against the oracle's SR image, identical
arg-max text priors; plus loss and clipped-gradient-norm agreement.  One oracle step costs seconds on the host cores."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tpgsr_oracle as O  # noqa: E402

DEV = "cuda"


def _threads():
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(n, 64)))


def _psnr(a, b):
    return float(O.calculate_psnr(a.detach().cpu().float(), b.detach().cpu().float()))


def _tpgsr(n_stu, seeds=(11, 12, 13)):
    from tpgsr_amd.model import tsrn
    from tpgsr_amd.model.crnn import crnn
    sd_sr = O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), seeds[0], tps_hw=(16, 64))
    sr = tsrn.TSRN_TL(STN=True, mask=True)
    sr.load_state_dict(sd_sr)
    sd_t = O.recipe_state_dict(O.crnn_spec(), seeds[1])
    teacher = crnn.CRNN(32, 1, 37, 256)
    teacher.load_state_dict(sd_t)
    stus, sd_s = [], []
    for k in range(n_stu):
        sd = O.recipe_state_dict(O.crnn_spec(), seeds[2] + k)
        s = crnn.CRNN(32, 1, 37, 256)
        s.load_state_dict(sd)
        stus.append(s.to(DEV).train())
        sd_s.append(sd)
    return sr.to(DEV).train(), stus, teacher.to(DEV).eval(), sd_sr, sd_s, sd_t


def test_c2_bs48_step0_vs_oracle():
    """BASELINE configs[1]: TSRN (STN + mask) fp32, bs 48 -- step 0 against oracle.tsrn_train_step on the same batch."""
    from tpgsr_amd.interfaces.super_resolution import TSRNTrainStep
    from tpgsr_amd.model import tsrn
    _threads()
    sd = O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True), 1234, tps_hw=(16, 64))
    net = tsrn.TSRN(STN=True, mask=True)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    lr, hr = O.synthetic_batch(48, 1234)
    ts = TSRNTrainStep(net)
    loss = ts.step(lr.to(DEV), hr.to(DEV))
    torch.cuda.synchronize()
    p = O.as_params(sd)
    opt = O.AdamState([p[k] for k in O.trainable_keys(p)])
    ref = O.tsrn_train_step(p, opt, lr, hr)
    gn = ts.opt.grad_norm(net).item()
    dpsnr = abs(_psnr(ts.last_sr, hr) - _psnr(ref["sr"], hr))
    print(f"C2 bs48: loss {loss.item():.6f} vs {ref['loss'].item():.6f}; grad norm {gn:.4f} vs {float(ref['grad_norm']):.4f}; "
          f"dPSNR {dpsnr:.2e} dB; SR max err {(ts.last_sr.cpu() - ref['sr']).abs().max().item():.2e}")
    assert abs(loss.item() - ref["loss"].item()) < 2e-4 * ref["loss"].item()
    assert abs(gn - float(ref["grad_norm"])) < 3e-3 * float(ref["grad_norm"])
    assert dpsnr < 1e-3
    # parameters after the Adam step: every element moved by at most lr; same direction as the oracle wherever the gradient is
    # clearly non-zero (Adam's first step is lr * sign(g) up to eps)
    flat = torch.cat([p[k].detach().reshape(-1) for k, _ in net.named_parameters()])
    mine = torch.cat([q.detach().reshape(-1).cpu() for _, q in net.named_parameters()])
    assert (mine - flat).abs().max() < 2.1e-3


def test_c3_bs48_step0_vs_oracle():
    """BASELINE configs[2] (= one rank of configs[3]): TSRN_TL + teacher + one student, stu_iter 1, bs 48."""
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    _threads()
    sr, stus, teacher, sd_sr, sd_s, sd_t = _tpgsr(1)
    lr, hr = O.synthetic_batch(48, 1234)
    ts = TPGSRTrainStep([sr], stus, teacher, stu_iter=1)
    loss = ts.step(lr.to(DEV), hr.to(DEV))
    torch.cuda.synchronize()
    ps, pt, pu = O.as_params(sd_sr), O.as_params(sd_t, False), [O.as_params(x) for x in sd_s]
    opt = O.AdamState([ps[k] for k in O.trainable_keys(ps)] + [q[k] for q in pu for k in O.trainable_keys(q)])
    ref = O.tpgsr_train_step([ps], pu, pt, opt, lr, hr, stu_iter=1)
    gn = ts.opt.grad_norm(sr).item()
    dpsnr = abs(_psnr(ts.last_sr, hr) - _psnr(ref["sr"], hr))
    am = ts.last_p.cpu().permute(1, 0, 2).argmax(-1)                 # (T, N)
    am_ref = ref["priors"][0].argmax(-1)
    print(f"C3 bs48: loss {loss.item():.6f} vs {ref['loss'].item():.6f}; SR grad norm {gn:.4f} vs {float(ref['grad_norms'][0]):.4f}; "
          f"dPSNR {dpsnr:.2e} dB; arg-max mismatches {(am != am_ref).sum().item()} / {am.numel()}")
    assert abs(loss.item() - ref["loss"].item()) < 3e-4 * ref["loss"].item()
    assert abs(gn - float(ref["grad_norms"][0])) < 5e-3 * float(ref["grad_norms"][0])
    assert dpsnr < 1e-3
    assert torch.equal(am, am_ref)
    assert (ts.last_p.cpu().permute(1, 0, 2) - ref["priors"][0]).abs().max() < 1e-5


def argmax_mismatches(p, p_ref):
    """(number of positions whose arg-max class differs, the oracle's top-1 - top-2 probability margin at the worst of them): a
    mismatch with a margin far above rounding noise is a real disagreement, one at a margin of 1e-7 a tie the oracle itself breaks by
    rounding -- printed so a failure says which it is"""
    bad = p.argmax(-1) != p_ref.argmax(-1)
    if not bool(bad.any()):
        return 0, 0.0
    top2 = p_ref.topk(2, -1).values
    return int(bad.sum()), float((top2[..., 0] - top2[..., 1])[bad].max())


def test_c5_shape_stu_iter3_sr_share_bs32_vs_oracle():
    """BASELINE configs[4] per rank: stu_iter 3, sr_share (one SR net, three forwards / backwards per step), three student
    recognisers, bs 32, STN on.  (The 'ASTER prior' of the config text has no reference implementation -- SURVEY 2a -- so
    the students are CRNNs, the only prior generator with an oracle.)"""
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    _threads()
    sr, stus, teacher, sd_sr, sd_s, sd_t = _tpgsr(3, seeds=(21, 22, 23))
    lr, hr = O.synthetic_batch(32, 555)
    ts = TPGSRTrainStep([sr], stus, teacher, stu_iter=3, sr_share=True, tpg_share=False)
    loss = ts.step(lr.to(DEV), hr.to(DEV))
    torch.cuda.synchronize()
    ps, pt, pu = O.as_params(sd_sr), O.as_params(sd_t, False), [O.as_params(x) for x in sd_s]
    opt = O.AdamState([ps[k] for k in O.trainable_keys(ps)] + [q[k] for q in pu for k in O.trainable_keys(q)])
    ref = O.tpgsr_train_step([ps], pu, pt, opt, lr, hr, stu_iter=3, sr_share=True, tpg_share=False)
    gn = ts.opt.grad_norm(sr).item()
    dpsnr = abs(_psnr(ts.last_sr, hr) - _psnr(ref["sr"], hr))
    mism, margins = [], []
    for i in range(3):                                               # every stage's prior
        m_i, worst_i = argmax_mismatches(ts._static["p"][i].cpu().permute(1, 0, 2), ref["priors"][i])
        mism.append(m_i)
        margins.append(worst_i)
    numel = ref["priors"][0].argmax(-1).numel()
    print(f"C5-shape bs32: loss {loss.item():.6f} vs {ref['loss'].item():.6f}; SR grad norm {gn:.4f} vs "
          f"{float(ref['grad_norms'][0]):.4f}; dPSNR {dpsnr:.2e} dB; arg-max mismatches per stage {mism} / {numel} "
          f"(oracle top-2 margin at the mismatches: {margins})")
    assert abs(loss.item() - ref["loss"].item()) < 5e-4 * ref["loss"].item()
    assert abs(gn - float(ref["grad_norms"][0])) < 2e-2 * float(ref["grad_norms"][0])
    assert dpsnr < 1e-3
    # north_star: IDENTICAL arg-max text priors -- in every stage, also the later ones that read the previous SR image through the
    # ill-conditioned TPS resampling.  (Rounds 1-4 allowed 0.5 % there; the measured count was always 0.)
    assert mism == [0, 0, 0], (mism, margins)
    # determinism of the cascade (the bicubic adjoint is a gather): a second replica gives bitwise the same step
    sr2, stus2, teacher2, *_ = _tpgsr(3, seeds=(21, 22, 23))
    ts2 = TPGSRTrainStep([sr2], stus2, teacher2, stu_iter=3, sr_share=True, tpg_share=False)
    loss2 = ts2.step(lr.to(DEV), hr.to(DEV))
    torch.cuda.synchronize()
    assert loss2.item() == loss.item()
    assert torch.equal(ts2.pool.flat, ts.pool.flat)


def test_module_api_two_forwards_before_backward():
    """fwd, fwd, bwd, bwd through the nn.Module API: every training forward owns a workspace slot until its
    backward ran, so the first backward sees its own activations."""
    from tpgsr_amd.model import tsrn
    sd = O.recipe_state_dict(O.tsrn_spec(STN=False, mask=True, srb_nums=2), 77)
    net = tsrn.TSRN(STN=False, mask=True, srb_nums=2)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    xa, _ = O.synthetic_batch(2, 1)
    xb, _ = O.synthetic_batch(2, 2)
    ga = torch.randn(2, 4, 32, 128, generator=torch.Generator().manual_seed(5)).to(DEV)
    gb = torch.randn(2, 4, 32, 128, generator=torch.Generator().manual_seed(6)).to(DEV)
    eng = net._engine()
    # sequential reference
    ya = net(xa.to(DEV)); (ya * ga).sum().backward()
    yb = net(xb.to(DEV)); (yb * gb).sum().backward()
    torch.cuda.synchronize()
    ref = eng.arena.grad.clone()
    # running statistics moved twice: reload to compare like with like
    net.load_state_dict(sd)
    eng.arena.grad.zero_()
    ya = net(xa.to(DEV))
    yb = net(xb.to(DEV))                      # second forward before the first backward
    (ya * ga).sum().backward()
    (yb * gb).sum().backward()
    torch.cuda.synchronize()
    assert len(eng._live) == 0
    assert (eng.arena.grad - ref).abs().max() <= 1e-6 * ref.abs().max()
    # more outstanding forwards than slots: the oldest one's backward must raise, not silently use overwritten activations
    outs = [net(xa.to(DEV)) for _ in range(eng.MAX_LIVE + 1)]
    with pytest.raises(RuntimeError, match="overwritten"):
        outs[0].sum().backward()
    outs[-1].sum().backward()
