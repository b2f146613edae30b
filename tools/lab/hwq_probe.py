#!/usr/bin/env python
"""Does `import tpgsr_amd` still get its GPU_MAX_HW_QUEUES=8 in when torch is imported first?  C3 step with the collectives forced at
world size 1 (the case that needs more than four hardware queues: 8.0 vs 7.2 ms in round 3) timed in three set-ups, one process each:
  env4 : GPU_MAX_HW_QUEUES=4 exported                      (the slow reference)
  pkg  : nothing exported, `import torch` THEN `import tpgsr_amd` (the package sets 8 before the first HIP call)
  late : nothing exported, HIP initialised (torch.cuda.init()) BEFORE the package is imported (round 5: measured to work as well)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1:
    mode = sys.argv[1]
    os.environ.pop("GPU_MAX_HW_QUEUES", None)
    if mode == "env4":
        os.environ["GPU_MAX_HW_QUEUES"] = "4"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    if mode == "late":
        torch.cuda.init()
        torch.zeros(1, device="cuda")
    sys.path.insert(0, ROOT)
    import tpgsr_amd  # noqa: F401
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    sys.argv = sys.argv[:1]
    import warnings
    warnings.simplefilter("always")
    # bench.py itself setdefaults the variable before importing torch: here torch is already in, so its line is a no-op for HIP
    import bench
    from tpgsr_amd import kernels as K
    from tpgsr_amd.utils.synthetic import synthetic_batch
    K.set_conv_prec("x2")
    ts, _ = bench.build_step("c3", dev, 1, None, force_collectives=True)
    lr, hr = synthetic_batch(48, 1234)
    lr, hr = lr.to(dev), hr.to(dev)
    for _ in range(10):
        ts.step(lr, hr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        ts.step(lr, hr)
    torch.cuda.synchronize()
    print(f"{mode}: GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} "
          f"{(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step (C3 x2, collectives forced at world size 1)", flush=True)
    dist.destroy_process_group()
    sys.exit(0)

for rep in range(2):
    for mode in ("env4", "pkg", "late"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], capture_output=True, text=True, timeout=280)
        lines = [ln for ln in (r.stdout + r.stderr).splitlines() if ln.startswith(mode + ":") or "GPU_MAX_HW_QUEUES" in ln]
        print("\n".join(lines[-3:]) if lines else f"{mode}: rc {r.returncode}\n{r.stderr[-800:]}")
