#!/usr/bin/env python
"""How many compute units does a CU-masked HIP stream really get?  Times the register-only MFMA probe (4096 workgroups,
runtime ~ 1 / #CUs) on streams created with different masks."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tpgsr_amd import _lib, kernels as K  # noqa: E402


def timed(stream, out, blocks=4096, iters=500):
    with torch.cuda.stream(stream):
        K.mfma_probe(out, blocks, iters)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.mfma_probe(out, blocks, iters)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def main():
    out = torch.zeros(4, device="cuda")
    base = timed(torch.cuda.current_stream(), out)
    print(f"unmasked: {base:8.1f} us")
    for spec in sys.argv[1:] or ["256", "128", "64", "128/2", "64/4", "32/8", "0xffffffff", "0xffffffff00000000"]:
        words = K.parse_cu_mask(spec)
        arr = (C.c_uint * len(words))(*words)
        raw = _lib.load().tpgsr_stream_create(arr, len(words))
        if not raw:
            print(spec, "-> create failed:", _lib.load().tpgsr_last_error().decode())
            continue
        st = torch.cuda.ExternalStream(raw)
        us = timed(st, out)
        print(f"mask {spec:>22s} ({sum(bin(w).count('1') for w in words):3d} bits): {us:8.1f} us  => ~{256 * base / us:5.1f} CUs")


if __name__ == "__main__":
    main()
