#!/usr/bin/env python
"""bench.py -- training throughput of the TPGSR-TSRN hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1], "C2"): TSRN (STN + mask, srb 5, hidden 32) fp32, batch 48 per GPU, 16x64 -> 32x128
synthetic crops, one FULL training step = forward + ImageLoss(gradient) + backward + [RCCL all-reduce of the flat
gradient arena] + clip_grad_norm_(0.25) + Adam -- all hand-written HIP kernels behind libtpgsr_hip.so.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `value` = whole-job img/s with inputs resident in HBM.  `roofline` is measured live
with HIP events around the dominant kernel's launches (the fp32-MFMA implicit-GEMM conv) on the stream they run on;
`cpu_baseline` times the CPU oracle (oracle/tpgsr_oracle.py, a port of the reference's step) on this box's host cores
(rank 0, N=1 only, bounded sample)."""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BATCH = 48
LR_HW = (16, 64)


def synthetic_batch(n, seed, device):
    """HR = U[0,1) RGB + luminance-threshold mask channel; LR = 2x average pool of HR with its own mask."""
    g = torch.Generator().manual_seed(seed)
    hr = torch.rand(n, 3, LR_HW[0] * 2, LR_HW[1] * 2, generator=g)
    lr = torch.nn.functional.avg_pool2d(hr, 2)

    def add_mask(img):
        lum = 0.299 * img[:, 0:1] + 0.587 * img[:, 1:2] + 0.114 * img[:, 2:3]
        return torch.cat([img, (lum <= lum.mean(dim=(1, 2, 3), keepdim=True)).float()], 1)

    return add_mask(lr).contiguous().to(device), add_mask(hr).contiguous().to(device)


def conv_roofline(eng, N, H, W, reps=5):
    """Replay only the conv_fwd launches (forward + data-gradient instances of the MFMA implicit-GEMM kernel) of one
    training step, bracketed by HIP events on the launch stream; algorithmic FLOPs = 2*M*K*Cout per launch."""
    from tpgsr_amd._lib import ConvArgs
    pl = eng.plans(N, H, W, True)
    ops = []
    flops = 0.0
    for plan in (pl["fwd"], pl["bwd"]):
        for name, fn, args, _sid in plan.ops:
            if name == "tpgsr_conv_fwd":
                a = args[0]._obj          # the ConvArgs struct behind the recorded ctypes.byref()
                flops += 2.0 * (a.N * a.OH * a.OW) * (a.KH * a.KW * a.Cin) * a.Cout
                ops.append((fn, args))
    s = torch.cuda.current_stream().cuda_stream
    for fn, args in ops:      # warm
        fn(*args, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for fn, args in ops:
            fn(*args, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return dict(launches=len(ops), flops_per_step=flops, ms_per_step=ms,
                avg_us_per_launch=1e3 * ms / len(ops), tflops=flops / (ms * 1e-3) / 1e12)


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


def cpu_baseline_subprocess(timeout_s=150):
    """Run the CPU-oracle timing in a child process with a hard timeout (a mis-sized thread pool must never stall the bench)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True, text=True,
                           timeout=timeout_s)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # timeout / parse failure: report, never hide
        return {"value": None, "unit": "img/s", "cores": None, "kind": "port", "sample": f"cpu baseline failed: {type(e).__name__}: {e}"[:200]}


def cpu_baseline(seconds_budget=20.0):
    from oracle import tpgsr_oracle as O
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncores, 64)))
    sd = O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True), 1234, tps_hw=LR_HW)
    p = O.as_params(sd)
    opt = O.AdamState([p[k] for k in O.trainable_keys(p)])
    lr, hr = O.synthetic_batch(BATCH, 1234)
    O.tsrn_train_step(p, opt, lr, hr)          # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        O.tsrn_train_step(p, opt, lr, hr)
        n += 1
        if time.perf_counter() - t0 > seconds_budget or n >= 8:
            break
    dt = time.perf_counter() - t0
    return {"value": round(BATCH * n / dt, 2), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} full C2 train steps (bs {BATCH}) of oracle/tpgsr_oracle.py on the host CPU, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as a captured hipGraph (default: plain launches on two HIP streams; measured faster "
                         "because ROCm executes the graph's fork/join branches serially)")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)   # old spelling of the default
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm

    from tpgsr_amd.model import tsrn
    from tpgsr_amd.interfaces.super_resolution import TSRNTrainStep
    from oracle import tpgsr_oracle as O  # weights-by-recipe only (no oracle compute in the timed path)

    torch.manual_seed(0)
    net = tsrn.TSRN(scale_factor=2, width=128, height=32, STN=True, srb_nums=5, mask=True, hidden_units=32)
    net.load_state_dict(O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True), 1234, tps_hw=LR_HW))
    net = net.to(dev).train()
    ts = TSRNTrainStep(net, gradient=True, loss_weight=(1.0, 1e-4), lr=1e-3, betas=(0.5, 0.999), max_norm=0.25,
                       process_group=pg, world_size=world)
    ts.broadcast_parameters(0)
    lr_img, hr_img = synthetic_batch(BATCH, 1234 + rank, dev)

    use_graph = args.graph and not args.no_graph
    _log(f"rank {rank}/{world}: model on {dev}, capturing={use_graph}")
    if use_graph:
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
        value = BATCH * world * args.steps / dt
        eng = net._engine()
        out = {
            "metric": "training img/s (16x64->32x128, bs=48/GPU), TSRN STN+mask fp32 full train step",
            "value": round(value, 1), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: TSRN (STN+mask, srb 5, hidden 32) fp32 train step: fwd + ImageLoss(gradient) + bwd + "
                                   "clip 0.25 + Adam", "batch_per_gpu": BATCH, "global_batch": BATCH * world,
                       "lr_hw": list(LR_HW), "hr_hw": [32, 128], "parallelism": f"dp{world}",
                       "launch": "hipGraph replay" if use_graph else "recorded plan, plain launches: main stream + weight-gradient stream",
                       "kernel_launches_per_step": len(eng.plans(BATCH, *LR_HW, True)["fwd"]) + len(eng.plans(BATCH, *LR_HW, True)["bwd"])},
            "final_loss": round(final_loss, 5),
        }
        _log(f"timed region done: {ms:.3f} ms/step")
        if not args.no_roofline:
            r = conv_roofline(eng, BATCH, *LR_HW)
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_conv.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            out["roofline"] = {"kernel": "conv_fwd_kernel<true> (fp32-MFMA implicit GEMM: all conv/linear fwd + dgrad launches)",
                               "bound": "mfma", "achieved": round(r["tflops"], 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(r["tflops"] / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                               "launches_per_step": r["launches"], "avg_us_per_launch": round(r["avg_us_per_launch"], 2),
                               "gflop_per_launch": round(r["flops_per_step"] / r["launches"] / 1e9, 4),
                               "share_of_step_ms": round(r["ms_per_step"], 4),
                               "measured_mfma_only_peak": round(mfma_probe_tflops(), 1)}
            # whole-step view against SURVEY 8d's algorithmic constants (58.8 MB, 5.5 GFLOP per image per C2 step)
            out["step_roofline"] = {"hbm_frac": round(value / world * 58.8e6 / 8.0e12, 4),
                                    "fp32_flop_frac": round(value / world * 5.5e9 / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4)}
        if world == 1 and not args.no_cpu_baseline:
            _log("cpu baseline (subprocess, bounded)")
            out["cpu_baseline"] = cpu_baseline_subprocess()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
