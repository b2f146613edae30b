#!/usr/bin/env python
"""What ONE rank's RCCL all-reduce costs (world size 1: the reduction is the identity) for the two gradient buckets of the C3 step --
explains the difference between `bench.py --force-collectives` and the plain step; says nothing about N > 1."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for mb in (1, 14, 33, 47):
    n = mb * (1 << 20) // 4
    t = torch.zeros(n, device=dev)
    for _ in range(3):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(20):
        dist.all_reduce(t)
    e1.record()
    host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    print(f"{mb:3d} MB: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us on the stream, {host * 1e6:6.1f} us of host time per call")
dist.destroy_process_group()
