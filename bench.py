#!/usr/bin/env python
"""bench.py -- training throughput of the TPGSR-TSRN hot path on MI355X (BASELINE.json metric).

Default workload = the configuration the metric is quoted on, BASELINE.json configs[2] ("C3", = the `north_star` target
"TPGSR-TSRN training step, bs=48/GPU, 16x64 -> 32x128"): TSRN_TL (STN + mask, srb 5, hidden 32) + frozen teacher CRNN +
one student CRNN (text prior, stu_iter 1), batch 48 per GPU, one FULL training step of
interfaces/super_resolution.py:295-424 = teacher forward, student forward, softmax / distill loss / prior + prior dropout,
SR forward, ImageLoss(gradient), backward through SR net and student, [RCCL all-reduce of the flat gradient buffer in two
buckets, the SR bucket overlapped with the student backward], clip_grad_norm_(0.25) on the SR net, one Adam over SR net +
student -- all hand-written HIP kernels behind libtpgsr_hip.so.  `--gpus N` runs the same step data-parallel (= C4).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c5]
    python bench.py --eval [--steps K] [--warmup W]         # throughput of the evaluation pass (one GPU), its own line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

  --config c2: TSRN without text prior (BASELINE configs[1], fp32)       --config c5: stu_iter 3, sr_share, bs 32 (configs[4])

Prints ONE JSON line (rank 0).  `value` = whole-job img/s with inputs resident in HBM.  `roofline` is measured live:
the dominant kernel family's launches (the MFMA implicit-GEMM convolution: every conv / linear forward, data-gradient and
weight-gradient launch of the step, taken from the recorded plans of all networks) are replayed between HIP events on the stream
they run on, shape by shape;
`traffic` comes from two rocprofv3 PMC passes over a few steps of the same workload (tools/pmc_step.sh; --no-traffic skips them).
`cpu_baseline` times the CPU oracle (oracle/tpgsr_oracle.py, a port of the reference's step pinned against the imported
reference) on this box's host cores -- rank 0, N=1 only, a bounded sample."""
import argparse
import json
import os
import sys
import time

# The step runs on three HIP streams; a gradient exchange adds RCCL's.  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4): with a fifth stream two of them share a queue and an event wait of one blocks the other -- measured at world size 1 with
# the collectives forced: 8.02 ms per step on four queues, 7.17 on eight (plain step 7.08 either way; profiles/r03z_collective_path_ab.md).
# Read when the HIP runtime initialises, so it is set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense (= the fp32 vector rate)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA
LR_HW = (16, 64)

# algorithmic bytes / FLOPs per image per training step (fp32 activations = 2x the bf16 figures)
CONFIGS = {
    "c2": dict(batch=48, stu_iter=1, tl=False, mb_per_img=58.8, gflop_per_img=5.5,
               name="C2: TSRN (STN+mask, srb 5, hidden 32) fp32 train step: fwd + ImageLoss(gradient) + bwd + clip 0.25 + Adam"),
    "c3": dict(batch=48, stu_iter=1, tl=True, mb_per_img=84.4, gflop_per_img=11.5,
               name="C3: TPGSR-TSRN = TSRN_TL (STN+mask) + CRNN teacher/student text prior, stu_iter 1, full train step "
                    "(teacher fwd, student fwd+bwd, distill loss, prior dropout, SR fwd+bwd, ImageLoss(gradient), clip 0.25, Adam)"),
    "c5": dict(batch=32, stu_iter=3, tl=True, mb_per_img=242.0, gflop_per_img=31.7,
               name="C5: TPGSR-TSRN multi-stage, stu_iter 3, sr_share, three CRNN students + frozen teacher, full train step"),
}
K_POLICY = "f32"
ARITH = {"f32": "fp32 operands on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), fp32 accumulate",
         "x3": "fp32-equivalent: fp32 operands split exactly into 3 bf16 terms, 6 bf16 MFMAs per product block, fp32 accumulate "
               "(csrc/conv_xbf.hip); everything else fp32",
         "x3b2": "forward passes fp32-equivalent (x3: split into 3 bf16 terms, 6 MFMAs per product block); backward GEMMs (data and weight "
                 "gradients) on the two-term split (3 MFMAs per product block); fp32 accumulate everywhere",
         "x2": "two-term split: every fp32 operand as the sum of TWO bf16 terms (16 significand bits), 3 bf16 MFMAs per product block, fp32 "
               "accumulate, in the SR network, in all backward GEMMs AND in the frozen teacher recogniser's forward (its softmax is only the "
               "soft distillation target); the STUDENT text-prior generator's forward stays fp32-equivalent (x3), so its arg-max prior is "
               "computed exactly as under x3",
         "bf16": "bf16 operands (RNE from fp32) on the bf16 matrix cores with fp32 accumulation in the SR network and all backward "
                 "GEMMs; the text-prior generator's forward stays fp32-equivalent (split operands) so arg-max priors are identical "
                 "to the fp32 oracle; activations / statistics / recurrences / losses / optimiser fp32"}


def synthetic_batch(n, seed, device):
    """HR = U[0,1) RGB + luminance-threshold mask channel; LR = 2x average pool of HR with its own mask."""
    from tpgsr_amd.utils.synthetic import synthetic_batch as sb
    lr, hr = sb(n, seed, lr_hw=LR_HW)
    return lr.to(device), hr.to(device)


OPT_ARGS = dict(Transformation="None", FeatureExtraction="ResNet", SequenceModeling="None", Prediction="CTC", num_fiducial=20,
                input_channel=1, output_channel=512, hidden_size=256, num_class=37)      # main.py:60-75's option set for `--tpg OPT`


def build_step(cfg_key, dev, world=1, pg=None, force_collectives=False, tpg="crnn"):
    """networks (weights by recipe: no pretrained files exist) + the train-step driver of the chosen configuration.
    Product code only: the CPU oracle is not needed to build or run the benchmarked step."""
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep, TSRNTrainStep
    from tpgsr_amd.model import tsrn
    from tpgsr_amd.model.crnn import crnn
    from tpgsr_amd.utils.synthetic import init_by_recipe
    cfg = CONFIGS[cfg_key]
    if not cfg["tl"]:
        net = init_by_recipe(tsrn.TSRN(scale_factor=2, width=128, height=32, STN=True, srb_nums=5, mask=True, hidden_units=32), 1234)
        net = net.to(dev).train()
        ts = TSRNTrainStep(net, gradient=True, loss_weight=(1.0, 1e-4), lr=1e-3, betas=(0.5, 0.999), max_norm=0.25,
                           process_group=pg, world_size=world, force_collectives=force_collectives)
        return ts, [net]
    sr = init_by_recipe(tsrn.TSRN_TL(scale_factor=2, width=128, height=32, STN=True, srb_nums=5, mask=True, hidden_units=32), 11)
    if tpg == "opt":      # `--tpg OPT`: the None-ResNet-None-CTC recogniser as teacher and students (interfaces/super_resolution.py:77-80)
        from tpgsr_amd.model.crnn import model as opt_model
        make = lambda: opt_model.Model(OPT_ARGS)
    else:
        make = lambda: crnn.CRNN(32, 1, 37, 256)
    teacher = init_by_recipe(make(), 12)
    students = [init_by_recipe(make(), 13 + k).to(dev).train() for k in range(cfg["stu_iter"])]
    ts = TPGSRTrainStep([sr.to(dev).train()], students, teacher.to(dev).eval(), stu_iter=cfg["stu_iter"], sr_share=True,
                        tpg_share=False, gradient=True, loss_weight=(1.0, 1e-4), lr=1e-3, betas=(0.5, 0.999), max_norm=0.25,
                        process_group=pg, world_size=world, force_collectives=force_collectives)
    return ts, [sr] + students + [teacher]


def module_api_bench(cfg_key, dev, images_lr, images_hr, steps=20, warmup=6, fused_optimizer=False):
    """One C3 step as a user of the REFERENCE writes it (interfaces/super_resolution.py:295-424, verbatim but for the imports): the drop-in
    modules driven by torch autograd, `clip_grad_norm_` and `torch.optim.Adam` -- tests/test_crnn_gpu.py::test_dropin_module_api_c3_step is
    the same loop against the reference's recorded numbers.  Same networks / weights recipe / batch as the fused step of this line.
    fused_optimizer: the two-line change INTEGRATION.md offers -- `tpgsr_amd.optim.FusedAdam([model, stu_model], ..., clip_modules=[model])`
    in place of torch.optim.Adam + clip_grad_norm_ (three launches per module over the flat arena instead of ~10 foreach kernels over
    361 tensors and their Python bookkeeping); everything else stays the reference's loop."""
    from tpgsr_amd.interfaces.super_resolution import parse_crnn_data
    from tpgsr_amd.loss.image_loss import ImageLoss
    from tpgsr_amd.loss.semantic_loss import SemanticLoss
    from tpgsr_amd.model import tsrn
    from tpgsr_amd.model.crnn import crnn
    from tpgsr_amd.utils.synthetic import init_by_recipe
    model = init_by_recipe(tsrn.TSRN_TL(scale_factor=2, width=128, height=32, STN=True, srb_nums=5, mask=True, hidden_units=32), 11).to(dev).train()
    aster = init_by_recipe(crnn.CRNN(32, 1, 37, 256), 12).to(dev).eval()           # the frozen teacher (the reference calls it `aster`)
    stu_model = init_by_recipe(crnn.CRNN(32, 1, 37, 256), 13).to(dev).train()
    for q in aster.parameters():
        q.requires_grad = False
    image_crit, sem_loss = ImageLoss(gradient=True, loss_weight=[1, 1e-4]), SemanticLoss()
    if fused_optimizer:
        from tpgsr_amd.optim import FusedAdam
        optimizer_G = FusedAdam([model, stu_model], lr=1e-3, betas=(0.5, 0.999), clip_modules=[model], max_norm=0.25)
    else:
        optimizer_G = torch.optim.Adam(list(model.parameters()) + list(stu_model.parameters()), lr=1e-3, betas=(0.5, 0.999))
    drop_vec = torch.ones(images_lr.shape[0]).float()
    drop_vec[:int(images_lr.shape[0] // 4)] = 0.
    drop_vec = drop_vec.to(dev).view(-1, 1, 1, 1)

    def loop_body():
        label_vecs_hr = torch.nn.functional.softmax(aster(parse_crnn_data(images_hr[:, :3, :, :])).detach(), -1)
        label_vecs_logits = stu_model(parse_crnn_data(images_lr[:, :3, :, :]))
        label_vecs = torch.nn.functional.softmax(label_vecs_logits, -1)
        label_vecs_final = label_vecs.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
        loss_recog_distill = sem_loss(label_vecs, label_vecs_hr) * 100
        label_vecs_final = label_vecs_final * drop_vec
        cascade_images = model(images_lr, label_vecs_final)
        loss_img = image_crit(cascade_images, images_hr).mean() * 100
        loss_im = loss_img + loss_recog_distill
        optimizer_G.zero_grad()
        loss_im.backward()
        if not fused_optimizer:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25)
        optimizer_G.step()
        return loss_im

    for _ in range(warmup):
        loop_body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = loop_body()
    t_sub = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    B = images_lr.shape[0]
    return {"workload": "the same C3 step as the reference's loop body on the drop-in nn.Modules (model(images_lr, prior), loss.backward(), "
                        "clip_grad_norm_, torch.optim.Adam.step): autograd hands the HIP plans over, ATen runs softmax / permute / the per-tensor Adam",
            "steps": steps, "ms_per_step": round(1e3 * dt / steps, 4), "value": round(B * steps / dt, 1), "unit": "img/s",
            "host_submission_ms_per_step": round(1e3 * t_sub / steps, 4), "final_loss": round(float(loss.item()), 5), "arithmetic_policy": K_POLICY}


def _exchange_note(ts, world, forced):
    if world == 1 and not forced:
        return None
    nb = len(ts._exchanger().bounds)
    note = ("one flat fp32 buffer, %d RCCL all-reduce buckets launched from the weight-gradient stream as their gradients become final "
            "(SR net under the text-prior generator's backward%s)" % (nb, "; the generator from conv3 on, 95.6 %, under the rest of its backward; "
                                                                      "1.5 MB at the end" if nb == 3 else ""))
    return note + ("; forced at world size 1 (the all-reduce is the identity)" if world == 1 else "")


PEAK_BY_TERMS = {0: FP32_MFMA_PEAK_TFLOPS,          # v_mfma_f32_32x32x2_f32
                 3: BF16_MFMA_PEAK_TFLOPS / 6.0,    # fp32-equivalent: six v_mfma_f32_32x32x16_bf16 per product block
                 2: BF16_MFMA_PEAK_TFLOPS / 3.0,    # two-term split: three MFMAs per product block
                 1: BF16_MFMA_PEAK_TFLOPS}          # bf16 operands
TERMS_NAME = {0: "fp32 MFMA (v_mfma_f32_32x32x2_f32)", 3: "x3: split operands, 6 x v_mfma_f32_32x32x16_bf16 per block (fp32-equivalent)",
              2: "x2: two-term split, 3 x v_mfma_f32_32x32x16_bf16 per block", 1: "bf16 operands, 1 x v_mfma_f32_32x32x16_bf16 per block"}


def conv_family(nets):
    """Every launch of the MFMA implicit-GEMM convolution family in ONE training step, from the recorded plans of all networks:
    forward / data-gradient launches (tpgsr_conv_fwd), weight-gradient launches (tpgsr_conv_wgrad: the launch includes the dy
    pre-split of the halo kernel) and the slab reduces that finish them (no FLOPs of their own, their time is charged to `wgrad`)."""
    items = []
    for net in nets:
        for pl in net._engine()._plans.values():
            for pname in ("pre", "fwd", "fwd_b", "bwd", "bwd_b"):
                if pname not in pl:
                    continue
                for oi, (name, fn, args, _sid) in enumerate(pl[pname].ops):
                    if name == "tpgsr_conv_wgrad_batch":
                        # several independent 1x1 weight gradients in one launch (the BiLSTM layers'): FLOPs / bytes of its items
                        ws = pl[pname].meta.get(oi)
                        if ws:
                            c0 = ws[0].c
                            fl = sum(2.0 * w.c.N * w.c.OH * w.c.OW * w.c.Cin * w.c.Cout for w in ws)
                            by = sum(4.0 * w.c.N * w.c.OH * w.c.OW * (w.c.Cin + w.c.Cout) for w in ws)
                            items.append(dict(kind="wgrad", terms=c0.terms, shape=(c0.N, c0.H, c0.W, c0.Cin, c0.Cout, 1, 1),
                                              label=f"{len(ws)} GEMMs of a BiLSTM layer in one launch", flops=fl, bytes=by, fn=fn, args=args))
                        continue
                    if name == "tpgsr_conv_fwd":
                        a = args[0]._obj          # the ConvArgs struct behind the recorded ctypes.byref()
                        kind, terms = ("dgrad" if pname.startswith("bwd") else "fwd"), (a.terms if (a.terms and a.wt_bf) else 0)
                    elif name == "tpgsr_conv_wgrad":
                        a = args[0]._obj.c
                        kind, terms = "wgrad", a.terms
                    elif name == "tpgsr_gru_wgrad":
                        # every weight gradient of one GruBlock in one launch (csrc/gru_wgrad.hip): dWc [Cin x 192] + 2 x dWhh [32 x 96],
                        # contracted over the pixels; algorithmic bytes = loader(x) + h + dgi + dghn once
                        a = args[0]._obj.c
                        M = a.N * a.H * a.W
                        items.append(dict(kind="wgrad", terms=a.terms, shape=(a.N, a.H, a.W, a.Cin, 192, 1, 1), label="GruBlock, all weights",
                                          flops=2.0 * M * (a.Cin * 192 + 2 * 32 * 96), bytes=4.0 * M * (a.Cin + 64 + 192 + 64), fn=fn, args=args))
                        continue
                    elif name in ("tpgsr_wgrad_reduce_program", "tpgsr_wgrad_reduce"):
                        items.append(dict(kind="wgrad", terms=None, shape=("slab reduce",), flops=0.0, bytes=0.0, fn=fn, args=args))
                        continue
                    else:
                        continue
                    M, K = a.N * a.OH * a.OW, a.KH * a.KW * a.Cin
                    # algorithmic bytes: the input once + the output once (fwd / dgrad); input + dy once (wgrad), fp32
                    nbytes = 4.0 * (a.N * a.H * a.W * a.Cin + M * a.Cout)
                    items.append(dict(kind=kind, terms=terms, shape=(a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW), flops=2.0 * M * K * a.Cout,
                                      bytes=nbytes, fn=fn, args=args))
    return items


def conv_roofline(nets, reps=5):
    """Replay the convolution family's launches of one training step, shape group by shape group, bracketed by HIP events on the
    stream they are launched on.  achieved = algorithmic FLOPs (2 M K Cout per launch, from the recorded geometry) / measured time;
    peak = what the same launches would need at the peak of their arithmetic class (x3: 2500/6, x2: 2500/3, bf16: 2500, fp32 MFMA
    157.3 TFLOP/s), i.e. frac = time at peak / measured time -- over the WHOLE family, weight gradients and their slab reduces included."""
    items = conv_family(nets)
    groups = {}
    for it in items:
        groups.setdefault((it["kind"], it["terms"], it["shape"], it.get("label", "")), []).append(it)
    s = torch.cuda.current_stream().cuda_stream
    rows = []
    for (kind, terms, shape, label), its in groups.items():
        for it in its:      # warm
            it["fn"](*it["args"], s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for it in its:
                it["fn"](*it["args"], s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = sum(it["flops"] for it in its)
        peak = PEAK_BY_TERMS.get(terms)
        rows.append(dict(kind=kind, terms=terms, shape=shape, label=label, launches=len(its), ms=ms, flops=flops, bytes=sum(it["bytes"] for it in its),
                         ms_at_peak=(flops / (peak * 1e12) * 1e3) if peak else 0.0))

    def agg(sel):
        r = [x for x in rows if sel(x)]
        ms, flops, at_peak = sum(x["ms"] for x in r), sum(x["flops"] for x in r), sum(x["ms_at_peak"] for x in r)
        n = sum(x["launches"] for x in r if x["terms"] is not None)
        return dict(launches=n, gflop=flops / 1e9, ms=ms, tflops=flops / (ms * 1e-3) / 1e12 if ms else 0.0,
                    peak=flops / (at_peak * 1e-3) / 1e12 if at_peak else 0.0, frac=at_peak / ms if ms else 0.0,
                    alg_bytes=sum(x["bytes"] for x in r))

    total = agg(lambda x: True)
    by_kind = {"fwd+dgrad": agg(lambda x: x["kind"] in ("fwd", "dgrad")), "wgrad (+ dy split + slab reduce)": agg(lambda x: x["kind"] == "wgrad")}
    by_terms = {TERMS_NAME[t]: agg(lambda x, t=t: x["terms"] == t) for t in sorted({x["terms"] for x in rows if x["terms"] is not None})}
    table = []
    for x in sorted(rows, key=lambda x: -x["ms"]):
        if x["terms"] is None:
            table.append(dict(kind="wgrad slab reduce", launches=x["launches"], us_per_launch=round(1e3 * x["ms"] / x["launches"], 1)))
            continue
        N_, H_, W_, Ci, Co, KH, KW = x["shape"]
        table.append(dict(kind=x["kind"], shape=f"N{N_} {H_}x{W_} {Ci}->{Co} {KH}x{KW}" + (f" ({x['label']})" if x["label"] else ""), terms=x["terms"], launches=x["launches"],
                          us_per_launch=round(1e3 * x["ms"] / x["launches"], 1), tflops=round(x["flops"] / (x["ms"] * 1e-3) / 1e12, 1),
                          frac=round(x["ms_at_peak"] / x["ms"], 3)))
    return dict(total=total, by_kind=by_kind, by_terms=by_terms, table=table)


def measure_traffic(cfg_key, timeout_s=400):
    """HBM bytes of one training step from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate
    rocprofv3 passes (--pmc with --kernel-trace only), each scaled by a calibration launch of known byte count taken in the same pass
    (FETCH_SIZE under-reports wide streaming reads by 2x on gfx950).  tools/pmc_step.sh; None when rocprofv3 is not there or fails."""
    import shutil
    import subprocess
    if not shutil.which("rocprofv3"):
        return None
    out = os.path.join("gpurun_out", "bench_pmc")
    result = os.path.join(ROOT, out, "traffic.json")
    if os.path.exists(result):        # a file left by an earlier run must never be reported as this run's measurement
        os.remove(result)
    try:
        subprocess.run(["bash", os.path.join(ROOT, "tools", "pmc_step.sh"), out, cfg_key, "4"], cwd=ROOT, capture_output=True, text=True,
                       timeout=timeout_s, env=dict(os.environ, GRAFT_REPO_ROOT=ROOT))
        return json.load(open(result))
    except Exception as e:   # report, never hide
        _log(f"traffic measurement failed: {type(e).__name__}: {e}")
        return None


def _log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def mfma_probe_tflops():
    """register-only v_mfma_f32_32x32x2_f32 loop: what this chip sustains at its real clock (datasheet: 157.3)"""
    from tpgsr_amd import kernels as K
    out = torch.zeros(4, device="cuda")
    blocks, iters = 4096, 1000
    K.mfma_probe(out, blocks, iters)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.mfma_probe(out, blocks, iters)
    e1.record()
    torch.cuda.synchronize()
    return blocks * 4 * 2.0 * iters * 4096 / (e0.elapsed_time(e1) * 1e-3) / 1e12


def cpu_baseline_subprocess(cfg_key, timeout_s=240):
    """Run the CPU-oracle timing in a child process with a hard timeout (a mis-sized thread pool must never stall the bench)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--config", cfg_key],
                           capture_output=True, text=True, timeout=timeout_s)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # timeout / parse failure: report, never hide
        return {"value": None, "unit": "img/s", "cores": None, "kind": "port", "sample": f"cpu baseline failed: {type(e).__name__}: {e}"[:200]}


def cpu_baseline(cfg_key, seconds_budget=25.0, max_steps=6):
    """The oracle's train step of the same configuration on the host cores (same batch size, same synthetic inputs)."""
    from oracle import tpgsr_oracle as O
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncores, 64)))
    cfg = CONFIGS[cfg_key]
    B = cfg["batch"]
    lr, hr = O.synthetic_batch(B, 1234)
    if not cfg["tl"]:
        p = O.as_params(O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True), 1234, tps_hw=LR_HW))
        opt = O.AdamState([p[k] for k in O.trainable_keys(p)])
        step = lambda: O.tsrn_train_step(p, opt, lr, hr)
    else:
        ps = O.as_params(O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 11, tps_hw=LR_HW))
        pt = O.as_params(O.recipe_state_dict(O.crnn_spec(), 12), False)
        pu = [O.as_params(O.recipe_state_dict(O.crnn_spec(), 13 + k)) for k in range(cfg["stu_iter"])]
        opt = O.AdamState([ps[k] for k in O.trainable_keys(ps)] + [q[k] for q in pu for k in O.trainable_keys(q)])
        step = lambda: O.tpgsr_train_step([ps], pu, pt, opt, lr, hr, stu_iter=cfg["stu_iter"], sr_share=True, tpg_share=False)
    step()          # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        if time.perf_counter() - t0 > seconds_budget or n >= max_steps:
            break
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 2), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} full {cfg_key.upper()} train steps (bs {B}) of oracle/tpgsr_oracle.py on the host CPU, {dt:.1f} s"}


def build_eval(dev, stu_iter=1):
    """the evaluation pass's networks (interfaces/super_resolution.py:540-900 / :1365-1432): eval-mode TSRN_TL + text-prior generator(s)
    + the evaluation recogniser (`--test_model CRNN`), weights by recipe"""
    from tpgsr_amd.interfaces.super_resolution import TextSREvaluator
    from tpgsr_amd.model import tsrn
    from tpgsr_amd.model.crnn import crnn
    from tpgsr_amd.utils.synthetic import init_by_recipe
    sr = init_by_recipe(tsrn.TSRN_TL(scale_factor=2, width=128, height=32, STN=True, srb_nums=5, mask=True, hidden_units=32), 11).to(dev).eval()
    tpgs = [init_by_recipe(crnn.CRNN(32, 1, 37, 256), 13 + k).to(dev).eval() for k in range(stu_iter)]
    rec = init_by_recipe(crnn.CRNN(32, 1, 37, 256), 12).to(dev).eval()
    ev = TextSREvaluator([sr], tpgs, recognizer=rec, stu_iter=stu_iter, sr_share=True, tpg_share=False)
    # one eval batch = each generator once, the SR network stu_iter times, the recogniser three times (SR / LR / HR strings)
    return ev, [sr] * stu_iter + tpgs + [rec] * 3


def eval_bench(dev, steps, warmup, roofline=True, cpu=True):
    """`bench.py --eval`: throughput of the EVALUATION pass (the reference's only printed throughput is this loop's `fps`,
    interfaces/super_resolution.py:1373-1427).  One step = TextSREvaluator.eval_batch on a resident batch of 48: bicubic gray resize ->
    text-prior generator -> softmax prior -> SR network (eval-mode BatchNorm folded into the consumers' loaders, STN off), PSNR + SSIM of
    SR vs HR, CRNN recognition of SR / LR / HR with on-device CTC greedy decoding, strings compared on the host."""
    from tpgsr_amd import kernels as K
    B = 48
    ev, nets = build_eval(dev)
    lr_img, hr_img = synthetic_batch(B, 1234, dev)
    labels = ["text%02d" % i for i in range(B)]
    for _ in range(warmup):
        out = ev.eval_batch(lr_img, hr_img, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = ev.eval_batch(lr_img, hr_img, labels)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the device part alone: the SR images only (what `sr_time` brackets in the reference's loop, without its missing synchronize)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        ev.super_resolve(lr_img)
    torch.cuda.synchronize()
    dt_sr = time.perf_counter() - t1
    res = {"metric": "evaluation img/s (16x64->32x128, bs=48), TPGSR-TSRN eval_batch: SR + PSNR/SSIM + CRNN strings of SR/LR/HR",
           "value": round(B * steps / dt, 1), "unit": "img/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": round(1e3 * dt / steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": {"f32": "f32", "x3": "bf16x3", "x3b2": "bf16x3", "x2": "bf16x2", "bf16": "bf16"}[K.POLICY], "arithmetic_policy": K.POLICY,
           "data": "synthetic",
           "config": {"workload": "TPGSR-TSRN evaluation pass (TextSREvaluator.eval_batch), stu_iter 1, eval-mode networks, STN off, CRNN "
                                  "evaluation recogniser, strings built on the host", "batch_per_gpu": B, "lr_hw": list(LR_HW), "hr_hw": [32, 128],
                      "arithmetic": ARITH[K.POLICY]},
           "super_resolve_only": {"ms_per_step": round(1e3 * dt_sr / steps, 4), "value": round(B * steps / dt_sr, 1), "unit": "img/s"},
           "psnr": float(out["psnr"]), "ssim": float(out["ssim"])}
    if roofline:
        t = conv_roofline(nets)
        tt = t["total"]
        res["roofline"] = {"kernel": "MFMA implicit-GEMM convolution family: every conv / linear forward launch of one evaluation batch (SR network, "
                                     "text-prior generator, recogniser x 3)", "bound": "mfma", "achieved": round(tt["tflops"], 2),
                           "peak": round(tt["peak"], 1), "unit": "TFLOP/s", "frac": round(tt["frac"], 4), "traffic": None,
                           "launches_per_step": tt["launches"], "gflop_per_step": round(tt["gflop"], 2),
                           "ms_per_step_replayed": round(tt["ms"], 4), "algorithmic_bytes_per_launch": round(tt["alg_bytes"] / max(1, tt["launches"])),
                           "per_shape": t["table"][:10]}
    if cpu:
        res["cpu_baseline"] = cpu_baseline_subprocess("eval")
    return res


def cpu_eval_baseline(seconds_budget=20.0, max_steps=40):
    """oracle/tpgsr_oracle.py's tpgsr_eval_step (a port of the reference's evaluation branch) on the host cores, same batch"""
    from oracle import tpgsr_oracle as O
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncores, 64)))
    B = 48
    lr, hr = O.synthetic_batch(B, 1234)
    ps = O.as_params(O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 11, tps_hw=LR_HW), False)
    pt = O.as_params(O.recipe_state_dict(O.crnn_spec(), 13), False)
    pr = O.as_params(O.recipe_state_dict(O.crnn_spec(), 12), False)
    with torch.no_grad():
        O.tpgsr_eval_step([ps], [pt], pr, lr, hr)
        t0 = time.perf_counter()
        n = 0
        while True:
            O.tpgsr_eval_step([ps], [pt], pr, lr, hr)
            n += 1
            if time.perf_counter() - t0 > seconds_budget or n >= max_steps:
                break
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 2), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} evaluation batches (bs {B}) of oracle/tpgsr_oracle.py tpgsr_eval_step on the host CPU, {dt:.1f} s"}


def main():
    # ONE JSON line on stdout and nothing else: libraries write there too (RCCL prints a version banner when the process exits, gloo its
    # connection messages), so the descriptor the line goes to is kept aside and fd 1 is pointed at stderr for everybody else
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS) + ["eval"], default="c3")
    ap.add_argument("--prec", choices=["f32", "x3", "x3b2", "x2", "bf16"], default=None,
                    help="arithmetic of the MFMA GEMMs, see tpgsr_amd/kernels.py (default: TPGSR_CONV_PREC if set, else x2 -- BASELINE.json quotes "
                         "this configuration in bf16; x2 = two bf16 terms per operand holds the north_star gates, tests/test_policy_x2_gpu.py)")
    ap.add_argument("--tpg", choices=["crnn", "opt"], default="crnn", help="text-prior generator of the TPGSR configurations (the reference's --tpg): "
                                                                           "CRNN (BASELINE's) or the OPT None-ResNet-None-CTC recogniser")
    ap.add_argument("--force-collectives", action="store_true", help="world size 1 with the RCCL gradient exchange forced on (diagnostic)")
    ap.add_argument("--alt-prec", default="x3", help="a second policy timed after the headline (same step, same batch) and printed as "
                                                      "`alt_precision` of the same line; 'none' skips it")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as a captured hipGraph (default: plain launches on several HIP streams; measured "
                         "faster because ROCm executes the graph's fork/join branches serially)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-module-api", action="store_true", help="skip the `module_api` leg (the reference's loop body on the drop-in modules: torch autograd + torch.optim.Adam)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 PMC passes behind roofline.traffic (~1 min)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--eval", action="store_true", help="time the EVALUATION pass instead (TextSREvaluator.eval_batch, bs 48, one GPU): its own "
                                                        "JSON line with its own roofline; the training line carries a short `eval` key either way")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_eval_baseline() if args.config == "eval" else cpu_baseline(args.config)), file=line_out, flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); the HIP path has no CPU fallback")
    # TPGSR_BENCH_SHARED_GPU=1 (self-test of the multi-rank code path on a one-GPU box, NOT a measurement): all ranks share
    # cuda:0 and talk over gloo
    shared = os.environ.get("TPGSR_BENCH_SHARED_GPU") == "1"
    dev_index = 0 if shared else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
    elif args.force_collectives:
        # ONE rank with the gradient exchange forced on: RCCL's all-reduce (the identity here) launched from inside the backward pass,
        # wait, average -- what the collective code path itself costs a step, measurable on a one-GPU box
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    if os.environ.get("TPGSR_BENCH_MAIN_PRIORITY"):     # experiment switch: the step's main stream at another HIP priority
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(os.environ["TPGSR_BENCH_MAIN_PRIORITY"])))
    from tpgsr_amd import kernels as K
    K.set_conv_prec(args.prec or os.environ.get("TPGSR_CONV_PREC") or "x2")
    global K_POLICY
    K_POLICY = K.POLICY
    if args.eval or args.config == "eval":
        if world > 1:
            raise SystemExit("bench.py --eval is a one-GPU measurement (the evaluation pass has no exchange step: N ranks = N replicas)")
        res = eval_bench(dev, args.steps, args.warmup, roofline=not args.no_roofline, cpu=not args.no_cpu_baseline)
        print(json.dumps(res), file=line_out, flush=True)
        return
    cfg = CONFIGS[args.config]
    B = cfg["batch"]
    torch.manual_seed(0)
    ts, nets = build_step(args.config, dev, world, pg, force_collectives=args.force_collectives, tpg=args.tpg)
    ts.broadcast_parameters(0)
    lr_img, hr_img = synthetic_batch(B, 1234 + rank, dev)

    _log(f"rank {rank}/{world}: {args.config} on {dev}, capturing={args.graph}")
    if args.graph:
        ts.capture(lr_img, hr_img, warmup=2)
        step = lambda: ts.replay()
    else:
        step = lambda: ts.step(lr_img, hr_img)

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    _log("warm-up done, timing")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_submit = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0
    _log(f"host submission {1e3 * t_submit / args.steps:.3f} ms/step, wall {1e3 * dt / args.steps:.3f} ms/step")
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    out = None
    if rank == 0:
        ms = 1e3 * dt / args.steps
        value = B * world * args.steps / dt
        # the step's recorded plans: kernel launches, and plan operations = launches + stream edges (fork / join / edge); the ~25 launches the
        # step makes outside plans (losses, softmax / prior, optimiser) are not in either count
        _plans = [pl[k] for m in nets for pl in m._engine()._plans.values() for k in ("pack", "pack_late", "pre", "fwd", "fwd_b", "bwd", "bwd_b") if k in pl]
        n_ops = sum(len(p.ops) for p in _plans)
        n_launch = sum(1 for p in _plans for op in p.ops if op[1] is not None)
        out = {
            "metric": "training img/s (16x64->32x128, bs=%d/GPU), %s full train step" % (B, "TPGSR-TSRN" if cfg["tl"] else "TSRN"),
            "value": round(value, 1), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "x3": "bf16x3", "x3b2": "bf16x3 fwd / bf16x2 bwd", "x2": "bf16x2", "bf16": "bf16"}[K_POLICY], "arithmetic_policy": K_POLICY, "data": "synthetic",
            "config": {"workload": cfg["name"] + ("" if args.tpg == "crnn" else " -- with the OPT (None-ResNet-None-CTC) recogniser as teacher and "
                                                  "student instead of CRNN: NOT BASELINE's configuration"), "tpg": args.tpg, "batch_per_gpu": B, "global_batch": B * world,
                       "lr_hw": list(LR_HW), "hr_hw": [32, 128], "parallelism": f"dp{world}",
                       "launch": "hipGraph replay" if args.graph else "recorded plans, plain launches: main + weight-gradient + teacher streams",
                       "kernel_launches_per_step": n_launch, "plan_ops_per_step": n_ops, "arithmetic": ARITH[K_POLICY],
                       "gradient_exchange": _exchange_note(ts, world, args.force_collectives)},
            "final_loss": round(final_loss, 5),
        }
        _log(f"timed region done: {ms:.3f} ms/step")
        if not args.no_roofline:
            r = conv_roofline(nets)
            t = r["total"]
            traffic = measure_traffic(args.config) if (world == 1 and not args.no_traffic) else None
            fam = ("conv fwd/dgrad (halo)", "conv fwd/dgrad (tile loop)", "conv wgrad", "dy_split", "wgrad slab reduce")
            fam_bytes = None
            if traffic:
                fam_bytes = sum(traffic[k]["by_class"].get(c, 0) for k in ("fetch", "write") for c in fam)

            def rnd(d):
                return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()}

            out["roofline"] = {"kernel": "MFMA implicit-GEMM convolution family: every conv / linear forward, data-gradient AND weight-gradient "
                                         "launch of the step (SR net, student and teacher recognisers), slab reduces and dy pre-splits charged to the weight gradients",
                               "bound": "mfma", "achieved": round(t["tflops"], 2), "peak": round(t["peak"], 1), "unit": "TFLOP/s",
                               "frac": round(t["frac"], 4),
                               "traffic": round(fam_bytes / t["launches"]) if fam_bytes else None,
                               "traffic_live": bool(fam_bytes),     # measured by THIS run's two PMC passes (a stale result file is deleted first)
                               "traffic_note": "HBM-side bytes per launch (mean over the family's launches): FETCH_SIZE + WRITE_SIZE of the family's kernels in one "
                                               "training step, two rocprofv3 --pmc passes, calibrated in-pass (tools/pmc_step.sh); algorithmic_bytes_per_launch beside it",
                               "algorithmic_bytes_per_launch": round(t["alg_bytes"] / t["launches"]),
                               "peak_note": "time-weighted over the arithmetic classes of the launches: fp32 MFMA 157.3; x3 2500/6 = 416.7; x2 2500/3 = 833.3; "
                                            "bf16 operands 2500 (dense bf16 MFMA peak, MI355X_MICROARCH.md)",
                               "launches_per_step": t["launches"], "gflop_per_step": round(t["gflop"], 2), "ms_per_step_replayed": round(t["ms"], 4),
                               "avg_us_per_launch": round(1e3 * t["ms"] / t["launches"], 2),
                               "by_kind": {k: rnd(v) for k, v in r["by_kind"].items()}, "by_class": {k: rnd(v) for k, v in r["by_terms"].items()},
                               "per_shape": r["table"][:16],
                               "measured_fp32_mfma_only_peak": round(mfma_probe_tflops(), 1)}
            # whole-step view against SURVEY 8d's algorithmic constants (fp32 bytes / FLOPs per image per step)
            out["step_roofline"] = {"hbm_frac": round(value / world * cfg["mb_per_img"] * 1e6 / 8.0e12, 4),
                                    "fp32_flop_frac": round(value / world * cfg["gflop_per_img"] * 1e9 / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4),
                                    "hbm_bytes_per_step_pmc": round(traffic["hbm_bytes_per_step"]) if traffic else None,
                                    "algorithmic_bytes_per_step": round(cfg["mb_per_img"] * 1e6 * B),
                                    "hbm_traffic_by_class_pmc": ({k: traffic[k]["by_class"] for k in ("fetch", "write")} if traffic else None)}
        if world == 1 and args.alt_prec not in ("none", K_POLICY):
            # the same step under a second arithmetic policy (fresh networks, same weights recipe, same batch), after the headline's
            # timed region: what the faster headline arithmetic buys against the fp32-equivalent one
            K.set_conv_prec(args.alt_prec)
            ts2, _nets2 = build_step(args.config, dev, world, pg)
            # fresh plans + first use of this policy's kernel instantiations (code-object loads) -- and the GPU has just idled through the
            # PMC passes' child processes: 40 steps (~0.3 s) bring its clocks back before the timed region (with 8, this line came out
            # ~0.35 ms per step above the same policy measured as the headline of its own run: 7.05 vs 6.69 ms, gpurun_out/r05x3)
            for _ in range(max(40, args.warmup)):
                ts2.step(lr_img, hr_img)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                ts2.step(lr_img, hr_img)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            out["alt_precision"] = {"arithmetic_policy": args.alt_prec, "arithmetic": ARITH[args.alt_prec], "steps": args.steps,
                                    "ms_per_step": round(1e3 * dt2 / args.steps, 4), "value": round(B * args.steps / dt2, 1), "unit": "img/s"}
            if not args.no_roofline:      # the same convolution-family replay under this policy (its launches, its class peaks)
                t2 = conv_roofline(_nets2)["total"]
                out["alt_precision"]["roofline"] = {"bound": "mfma", "achieved": round(t2["tflops"], 2), "peak": round(t2["peak"], 1),
                                                    "unit": "TFLOP/s", "frac": round(t2["frac"], 4), "launches_per_step": t2["launches"],
                                                    "ms_per_step_replayed": round(t2["ms"], 4)}
            K.set_conv_prec(K_POLICY)
            del ts2, _nets2
        if world == 1 and not args.no_roofline:
            # the evaluation pass of the same networks (north_star: "training/inference path"): a short measurement, the full line with its
            # own roofline is `bench.py --eval`
            e = eval_bench(dev, 30, 8, roofline=False, cpu=False)
            out["eval"] = {k: e[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "super_resolve_only")}
        if world == 1 and not args.no_roofline and cfg["tl"] and args.tpg == "crnn":
            # the same step with `--tpg OPT` (recorded plans traced from the operator-level network, tpgsr_amd/engine_functional.py)
            try:
                ts3, _n3 = build_step(args.config, dev, world, pg, tpg="opt")
                for _ in range(6):
                    ts3.step(lr_img, hr_img)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                for _ in range(20):
                    ts3.step(lr_img, hr_img)
                torch.cuda.synchronize()
                dt3 = time.perf_counter() - t3
                out["tpg_opt"] = {"workload": "the same step with the OPT recogniser (29 convolutions, ResNet) as teacher and student", "steps": 20,
                                  "ms_per_step": round(1e3 * dt3 / 20, 4), "value": round(B * 20 / dt3, 1), "unit": "img/s"}
                del ts3, _n3
            except Exception as e:      # reported, never hidden
                out["tpg_opt"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and cfg["tl"] and cfg["stu_iter"] == 1 and args.tpg == "crnn" and not args.no_module_api:
            # THE DROP-IN PATH: the reference's own loop body (interfaces/super_resolution.py:295-424) on the module API -- model(images_lr, prior),
            # loss.backward() through torch autograd, clip_grad_norm_, torch.optim.Adam.step -- instead of the fused TPGSRTrainStep above
            try:
                out["module_api"] = module_api_bench(args.config, dev, lr_img, hr_img, steps=20, warmup=6)
                out["module_api"]["vs_fused_step"] = round(out["module_api"]["ms_per_step"] / ms, 3)
                fo = module_api_bench(args.config, dev, lr_img, hr_img, steps=20, warmup=6, fused_optimizer=True)
                out["module_api"]["with_tpgsr_amd_FusedAdam"] = {k: fo[k] for k in ("ms_per_step", "value", "unit", "host_submission_ms_per_step", "final_loss")}
            except Exception as e:      # reported, never hidden
                out["module_api"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            _log("cpu baseline (subprocess, bounded)")
            out["cpu_baseline"] = cpu_baseline_subprocess(args.config)
        print(json.dumps(out), file=line_out, flush=True)
    if world > 1:
        torch.distributed.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
