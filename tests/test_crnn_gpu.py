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
NOISE = 1.5     # see tests/test_tsrn_gpu.py


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
    # adjoint (gather form, deterministic) vs autograd: the LR geometry (up-sampling) and the SR geometry of the later
    # cascade stages (32x128 -> 32x100, the only place the training step runs it), plus odd sizes that hit the border clamps
    for (n, c, h, w) in [(2, 4, 16, 64), (3, 4, 32, 128), (1, 3, 7, 11), (2, 3, 40, 250)]:
        x = torch.rand(n, c, h, w)
        xr = x.clone().requires_grad_(True)
        y = O.parse_crnn_data(xr)
        gy = torch.randn(y.shape)
        y.backward(gy)
        din = torch.full((n, c, h, w), 7.0, device=DEV)            # the kernel owns every element (no pre-zeroing)
        din2 = torch.full((n, c, h, w), -3.0, device=DEV)
        gyd = gy.to(DEV).contiguous()
        K.bicubic_gray_bwd(gyd, n, c, h, w, 32, 100, din)
        K.bicubic_gray_bwd(gyd, n, c, h, w, 32, 100, din2)
        torch.cuda.synchronize()
        assert (din.cpu() - xr.grad).abs().max() < 2e-5, (n, c, h, w)
        assert torch.equal(din, din2)                              # no atomics: bitwise reproducible


def test_semantic_loss_module_on_probabilities(golden_dir):
    """SemanticLoss.forward / backward as kernels on probability tensors, incl. rows that do NOT sum to 1"""
    from tpgsr_amd.loss.semantic_loss import SemanticLoss
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    p_ref, q = torch.tensor(g["p"]), torch.tensor(g["q"])
    for scale in (1.0, 0.7):
        pr = (p_ref * scale).clone().requires_grad_(True)
        ref = O.semantic_loss(pr, q)
        ref.backward()
        pd = (p_ref * scale).to(DEV).requires_grad_(True)
        out = SemanticLoss()(pd, q.to(DEV))
        (out * 3.0).backward()
        assert abs(out.item() - ref.item()) < 2e-6 * abs(ref.item())
        assert (pd.grad.cpu() / 3.0 - pr.grad).abs().max() < 1e-5 * pr.grad.abs().max()


@pytest.mark.parametrize("hw", [(4, 9), (6, 10), (5, 9)])      # even H (and W): the 2x2 backward fast paths; (5, 9): the gather form
@pytest.mark.parametrize("cfg", [((2, 2), (2, 2), (0, 0)), ((2, 2), (2, 1), (0, 1))])
def test_pool2d(cfg, hw):
    from tpgsr_amd import kernels as K
    k, s, pd = cfg
    N, (H, W), C = 2, hw, 32
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


@pytest.mark.parametrize("N,Kd,Nc,S", [(48, 256, 1024, 8), (48, 1024, 256, 32), (5, 64, 128, 1), (64, 96, 64, 2)])
def test_lstm_rec_gemm_and_step_vs_explicit_cell(N, Kd, Nc, S):
    """Split-K recurrent projection (both directions in one launch): slabs summed in order == A_d @ B_d; rows are strided
    views into a [N][T][...] sequence tensor exactly as the engine passes them.  Then one LSTM step (gates from the slabs)
    against the gate equations of nn.LSTM written out (model/crnn/crnn.py:10)."""
    from tpgsr_amd import kernels as K
    g = torch.Generator().manual_seed(N + Kd)
    T, t0, t1 = 5, 1, 3
    seq = torch.randn(N, T, 2, Kd, generator=g).to(DEV)           # direction d, time t_d: row n at seq[n, t_d, d]
    B = (torch.randn(2, Kd, Nc, generator=g) / Kd ** 0.5).to(DEV)
    out = torch.full((S, 2, N, Nc), float("nan"), device=DEV)
    a0 = seq.data_ptr() + 4 * ((t0 * 2 + 0) * Kd)
    a1 = seq.data_ptr() + 4 * ((t1 * 2 + 1) * Kd)
    K.lstm_rec_gemm(a0, a1, T * 2 * Kd, B[0], B[1], N, Kd, Nc, S, out)
    torch.cuda.synchronize()
    ref = torch.stack([seq[:, t0, 0].double().cpu() @ B[0].double().cpu(), seq[:, t1, 1].double().cpu() @ B[1].double().cpu()])
    got = out.double().cpu().sum(0)
    assert not torch.isnan(out).any()
    assert (got - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())
    if Nc % 4 or Nc // 4 * 4 != Nc or Kd * 4 != Nc:
        return
    # one forward LSTM step (s = 1 of T = 2) fed by the slabs: Hh = Kd, 4*Hh = Nc
    Hh, T2 = Kd, 2
    G = torch.randn(N, T2, 2, 4 * Hh, generator=g).to(DEV)        # input projections (+ b_ih) of both directions
    bhh = torch.randn(2, 4 * Hh, generator=g).to(DEV)
    Cst = torch.zeros(N, T2, 2, Hh, device=DEV)
    hs = torch.zeros(N, T2, 2 * Hh, device=DEV)
    Gref = G.double().cpu().clone()
    K.lstm_step_fwd(G, None, 0, bhh, Cst, hs, N, T2, Hh, 0)       # step 0: h_prev = 0
    Whh = (torch.randn(2, Hh, 4 * Hh, generator=g) / Hh ** 0.5).to(DEV)      # [d][K = Hh][4Hh] = W_hh^T
    gh = torch.empty(S, 2, N, 4 * Hh, device=DEV)
    prev = [hs.data_ptr() + 4 * ((0 if d == 0 else 1) * 2 * Hh + d * Hh) for d in range(2)]   # h_prev: t = 0 (fwd), t = 1 (reverse)
    K.lstm_rec_gemm(prev[0], prev[1], T2 * 2 * Hh, Whh[0], Whh[1], N, Hh, 4 * Hh, S, gh)
    K.lstm_step_fwd(G, gh, S, bhh, Cst, hs, N, T2, Hh, 1)
    torch.cuda.synchronize()

    def cell(pre, cprev):
        i, f, gg, o = pre.split(Hh, -1)
        c = torch.sigmoid(f) * cprev + torch.sigmoid(i) * torch.tanh(gg)
        return torch.sigmoid(o) * torch.tanh(c), c

    b, W = bhh.double().cpu(), Whh.double().cpu()
    for d in range(2):
        tA, tB = (0, 1) if d == 0 else (1, 0)                      # first / second time index processed by direction d
        h0, c0 = cell(Gref[:, tA, d] + b[d], torch.zeros(N, Hh, dtype=torch.float64))
        h1, c1 = cell(Gref[:, tB, d] + b[d] + h0 @ W[d], c0)
        assert (hs[:, tA, d * Hh:(d + 1) * Hh].double().cpu() - h0).abs().max() < 2e-6
        assert (hs[:, tB, d * Hh:(d + 1) * Hh].double().cpu() - h1).abs().max() < 5e-6
        assert (Cst[:, tB, d].double().cpu() - c1).abs().max() < 5e-6


def test_crnn_vs_golden(golden_dir, golden_policy):
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
    # Two tiers.  Everything ABOVE the last max-pool (conv6, both BiLSTMs) must agree at the fp32 level.  BELOW a pool the
    # gradient is a discontinuous function of the forward values: one pooling window whose two largest inputs differ by less than
    # the forward's fp32 rounding noise picks the other position in one of the two implementations, which reroutes that window's
    # whole gradient (tools/lab/xbf_diverge.py shows it: with identical arithmetic the same layers agree to 1e-6, and changing
    # only the accumulation ORDER of a conv -- fp32 matrix cores vs split bf16, tile loop vs halo kernel -- moves the figure
    # between 3e-3 and 9e-3 on this 3-image batch).  Those layers get the looser bound; the full-size tests (test_fullsize_gpu.py)
    # hold the training-step gates that matter (loss, gradient norm, PSNR, arg-max priors).
    above = ("cnn.conv6", "cnn.batchnorm6", "rnn.")          # the last pool follows conv5
    bad, worst_above, worst_below = [], 0.0, 0.0
    for n, q in net.named_parameters():
        ref = p[n].grad
        rel = (q.grad.cpu() - ref).norm().item() / max(ref.norm().item(), 1e-3 * gmax)
        top = n.startswith(above)
        worst_above, worst_below = (max(worst_above, rel), worst_below) if top else (worst_above, max(worst_below, rel))
        if rel > (3e-3 if top else 2e-2):
            bad.append((n, rel))
    print(f"crnn grads: worst above the last pool {worst_above:.2e}, below {worst_below:.2e}")
    assert not bad, bad[:10]
    rel = (gd.grad.cpu() - gr.grad).norm().item() / gr.grad.norm().item()
    print("dgray rel err", rel)
    assert rel < 2e-2


def _c3_models(seeds=(301, 302, 303), stn=True, n_sr=1, n_stu=1):
    from tpgsr_amd.model import tsrn
    from tpgsr_amd.model.crnn import crnn
    srs, sds = [], []
    for k in range(n_sr):
        sd = O.recipe_state_dict(O.tsrn_spec(STN=stn, mask=True, text_prior=True), seeds[0] + 10 * k, tps_hw=(16, 64))
        m = tsrn.TSRN_TL(STN=stn, mask=True); m.load_state_dict(sd); srs.append(m.to(DEV).train()); sds.append(sd)
    sd_t = O.recipe_state_dict(O.crnn_spec(), seeds[1])
    teacher = crnn.CRNN(32, 1, 37, 256); teacher.load_state_dict(sd_t); teacher = teacher.to(DEV).eval()
    stus, sd_s = [], []
    for k in range(n_stu):
        sd = O.recipe_state_dict(O.crnn_spec(), seeds[2] + 10 * k)
        s = crnn.CRNN(32, 1, 37, 256); s.load_state_dict(sd); stus.append(s.to(DEV).train()); sd_s.append(sd)
    return srs, stus, teacher, sds, sd_s, sd_t


def test_train_c3_step_vs_golden(golden_dir, golden_policy):
    """C3: TSRN_TL + teacher CRNN + one student, stu_iter 1 -- loss / grad-norm / arg-max prior vs the reference's numbers"""
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    t = np.load(os.path.join(golden_dir, "train_c3.npz"))
    srs, stus, teacher, *_ = _c3_models()
    ts = TPGSRTrainStep(srs, stus, teacher, stu_iter=1)
    assert ts.precision == golden_policy.name
    lr, hr = torch.tensor(t["lr"]).to(DEV), torch.tensor(t["hr"]).to(DEV)
    loss = ts.step(lr, hr)
    gn = ts.opt.grad_norm(srs[0])
    print("C3 step0", golden_policy.name, ts.precision, loss.item(), t["loss"][0], gn.item(), t["gnorm"][0])
    assert abs(loss.item() - t["loss"][0]) < golden_policy.tol(3e-4) * t["loss"][0]
    assert abs(gn.item() - t["gnorm"][0]) < 3e-3 * t["gnorm"][0]
    assert (ts.last_p.cpu().permute(1, 0, 2).argmax(-1).numpy() == t["prior_argmax_step0"]).all()
    l1 = ts.step(lr, hr).item()
    print("C3 step1", l1, t["loss"][1])
    assert abs(l1 - t["loss"][1]) < 2e-2 * t["loss"][1]


def test_train_c3_hipgraph_replay_equals_eager(golden_dir):
    """the whole TPGSR step -- three streams (SR prologue on the side stream, teacher + STN-head backward on the leaf stream), the
    persistent BiLSTM kernels with their in-kernel hand-offs, deferred join -- captured into ONE hipGraph and replayed: bitwise the
    eager step sequence (same kernels, same order, deterministic reductions)"""
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    t = np.load(os.path.join(golden_dir, "train_c3.npz"))
    lr, hr = torch.tensor(t["lr"]).to(DEV), torch.tensor(t["hr"]).to(DEV)
    (sa, ua, ta_, *_), (sb, ub, tb_, *_) = _c3_models(), _c3_models()
    ea, eb = TPGSRTrainStep(sa, ua, ta_, stu_iter=1), TPGSRTrainStep(sb, ub, tb_, stu_iter=1)
    eb.capture(lr, hr, warmup=1)          # one eager warm-up step has been applied; the capture itself executes nothing
    la = [ea.step(lr, hr).item() for _ in range(3)]
    lb = [eb.replay().item() for _ in range(2)]
    torch.cuda.synchronize()
    print(la, lb)
    assert lb[0] == la[1] and lb[1] == la[2]
    assert torch.equal(ea.pool.flat, eb.pool.flat)


def test_train_c3_two_independent_runs_bitwise(golden_dir):
    """the step is a fixed dependency graph over three streams with deterministic reductions: two independent builds of it stepping the
    same batch four times end in bitwise the same losses and parameters (a missing edge between streams would show up here)"""
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    t = np.load(os.path.join(golden_dir, "train_c3.npz"))
    lr, hr = torch.tensor(t["lr"]).to(DEV), torch.tensor(t["hr"]).to(DEV)
    (sa, ua, ta_, *_), (sb, ub, tb_, *_) = _c3_models(), _c3_models()
    ea, eb = TPGSRTrainStep(sa, ua, ta_, stu_iter=1), TPGSRTrainStep(sb, ub, tb_, stu_iter=1)
    la = [ea.step(lr, hr).item() for _ in range(4)]
    lb = [eb.step(lr, hr).item() for _ in range(4)]
    torch.cuda.synchronize()
    assert la == lb, (la, lb)
    assert torch.equal(ea.pool.flat, eb.pool.flat)


def test_cascade_two_stages_vs_oracle():
    """stu_iter 2, sr_share, two students, no STN: every gradient (SR net accumulated over both stages, both students,
    incl. the path through parse_crnn_data of stage 2) against oracle autograd, via one fused step's Adam-free grads.

    Conditioning, measured with the oracle itself in fp32 vs fp64 on exactly this case (tools/dbg/dbg_cascade.py and the
    fp64 probe quoted in DESIGN.md): the students' parameter gradients are ill-conditioned in fp32 -- the oracle's own
    fp32 result is 1.07e-2 / 1.22e-2 (relative, student 0 / 1) away from its fp64 result, the stage-1 prior gradient
    1.1e-2 -- so student tolerances are percent-level by nature (the HIP path lands at 1.09e-2 / 1.25e-2, i.e. at the
    oracle's own noise level); the SR-net gradients and the last-stage prior gradient (2.6e-5) are tight."""
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    srs, stus, teacher, sds, sd_s, sd_t = _c3_models(stn=False, n_sr=1, n_stu=2)
    lr, hr = O.synthetic_batch(4, 77)
    ps = O.as_params(sds[0]); pt = O.as_params(sd_t, False); pu = [O.as_params(x) for x in sd_s]
    opt = O.AdamState([ps[k] for k in O.trainable_keys(ps)] + [q[k] for q in pu for k in O.trainable_keys(q)])
    ref = O.tpgsr_train_step([ps], pu, pt, opt, lr, hr, stu_iter=2, sr_share=True, tpg_share=False, stn=False)
    ts = TPGSRTrainStep(srs, stus, teacher, stu_iter=2, sr_share=True, tpg_share=False)
    ts.pool.bind(torch.device(DEV, 0))
    teacher._engine().bind(torch.device(DEV, 0))
    loss = ts._phase_a(lr.to(DEV), hr.to(DEV))            # forward + backward only (no optimiser): raw gradients
    torch.cuda.synchronize()
    assert abs(loss.item() - ref["loss"].item()) < 3e-4 * ref["loss"].item()
    # oracle grads were clipped in place for the SR group: compare directions via the un-clipped norm ratio
    flat_ref = ref["grads"]
    n_sr = len(O.trainable_keys(ps))
    names_sr = O.trainable_keys(ps)
    coef = min(1.0, 0.25 / (float(ref["grad_norms"][0]) + 1e-6))
    P = dict(srs[0].named_parameters())
    gmax = max(g.norm().item() for g in flat_ref[:n_sr]) / coef
    bad, num, den = [], 0.0, 0.0
    for k, gref in zip(names_sr, flat_ref[:n_sr]):
        d = (P[k].grad.cpu() * coef - gref)
        num += d.double().pow(2).sum().item(); den += gref.double().pow(2).sum().item()
        rel = d.norm().item() / max(gref.norm().item(), 1e-3 * gmax * coef)
        if rel > 2e-2:
            bad.append((k, rel))
    print("cascade SR grads: global rel err", (num / den) ** 0.5, "worst", max([0] + [b[1] for b in bad]))
    # two stage gradients of similar size partly cancel in the shared SR net, which amplifies relative fp32 noise
    assert (num / den) ** 0.5 < 5e-3 and not bad, bad[:8]
    ofs = n_sr
    for j, q in enumerate(pu):
        Ps = dict(stus[j].named_parameters())
        keys = O.trainable_keys(q)
        gm = max(g.norm().item() for g in flat_ref[ofs:ofs + len(keys)])
        num = den = 0.0
        worst = (None, 0.0)
        for k, gref in zip(keys, flat_ref[ofs:ofs + len(keys)]):
            d = Ps[k].grad.cpu() - gref
            num += d.double().pow(2).sum().item(); den += gref.double().pow(2).sum().item()
            rel = d.norm().item() / max(gref.norm().item(), 1e-3 * gm)
            if rel > worst[1]:
                worst = (k, rel)
        print(f"cascade student {j}: global rel err {(num / den) ** 0.5:.3e}, worst per-tensor {worst}")
        assert (num / den) ** 0.5 < 3e-2, (j, (num / den) ** 0.5, worst)
        ofs += len(keys)


@pytest.mark.parametrize("optimizer", ["torch.optim.Adam", "tpgsr_amd.optim.FusedAdam"])
def test_dropin_module_api_c3_step(golden_dir, optimizer):
    """The reference's own loop body (interfaces/super_resolution.py:295-424) written against the drop-in modules:
    torch autograd + torch.optim.Adam + clip_grad_norm_ drive the HIP plans through the nn.Module API.  Second case: the two-line change
    INTEGRATION.md offers for the loop's host time -- FusedAdam over the modules' flat arenas (clip included) in place of the two torch calls."""
    fused = optimizer.endswith("FusedAdam")
    from tpgsr_amd.interfaces.super_resolution import parse_crnn_data
    from tpgsr_amd.loss.image_loss import ImageLoss
    from tpgsr_amd.loss.semantic_loss import SemanticLoss
    t = np.load(os.path.join(golden_dir, "train_c3.npz"))
    srs, stus, teacher, *_ = _c3_models()
    model, stu = srs[0], stus[0]
    for q in teacher.parameters():
        q.requires_grad = False
    image_crit, sem_loss = ImageLoss(gradient=True, loss_weight=[1, 1e-4]), SemanticLoss()
    if fused:
        from tpgsr_amd.optim import FusedAdam
        optimizer_G = FusedAdam([model, stu], lr=1e-3, betas=(0.5, 0.999), clip_modules=[model], max_norm=0.25)
    else:
        optimizer_G = torch.optim.Adam(list(model.parameters()) + list(stu.parameters()), lr=1e-3, betas=(0.5, 0.999))
    images_lr, images_hr = torch.tensor(t["lr"]).to(DEV), torch.tensor(t["hr"]).to(DEV)
    losses = []
    for step in range(2):
        label_vecs_hr = torch.nn.functional.softmax(teacher(parse_crnn_data(images_hr[:, :3, :, :])).detach(), -1)
        label_vecs_logits = stu(parse_crnn_data(images_lr[:, :3, :, :]))
        label_vecs = torch.nn.functional.softmax(label_vecs_logits, -1)
        label_vecs_final = label_vecs.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
        loss_recog_distill = sem_loss(label_vecs, label_vecs_hr) * 100
        drop_vec = torch.ones(images_lr.shape[0]).float()
        drop_vec[:int(images_lr.shape[0] // 4)] = 0.
        label_vecs_final = label_vecs_final * drop_vec.to(DEV).view(-1, 1, 1, 1)
        cascade_images = model(images_lr, label_vecs_final)
        loss_img = image_crit(cascade_images, images_hr).mean() * 100
        loss_im = loss_img + loss_recog_distill
        optimizer_G.zero_grad()
        loss_im.backward()
        if fused:
            optimizer_G.step()
            gn = optimizer_G.grad_norm(model)
        else:
            gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25)
            optimizer_G.step()
        losses.append(loss_im.item())
        if step == 0:
            assert abs(loss_im.item() - t["loss"][0]) < 3e-4 * t["loss"][0]
            assert abs(float(gn) - t["gnorm"][0]) < 3e-3 * t["gnorm"][0]
            assert (label_vecs.detach().argmax(-1).cpu().numpy() == t["prior_argmax_step0"]).all()
    print("drop-in loop losses", losses, t["loss"][:2])
    assert abs(losses[1] - t["loss"][1]) < 2e-2 * t["loss"][1]
    # state_dict round trip keeps the reference's key layout
    sd = model.state_dict()
    import json
    ref = json.load(open(os.path.join(golden_dir, "state_dict_layouts.json")))["tsrn_tl_stn_mask"]
    assert [(k, list(v.shape)) for k, v in sd.items()] == [(a, b) for a, b in ref]
