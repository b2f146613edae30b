#!/usr/bin/env python
"""time line of the whole-CU halo kernel (tpgsr_halo3_trace) next to the two-workgroup one (tpgsr_halo_trace): per-item stamps of a
producer and a consumer wave of workgroups 0 / 1.   usage: halo3_trace.py N H W Cin Cout KH KW pad terms"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tpgsr_amd import _lib, kernels as K  # noqa: E402

N, H, W, Ci, Co, KH, KW, pad, terms = (int(v) for v in sys.argv[1:10])
g = K.ConvGeom(N, H, W, Ci, Co, KH, KW, pad, pad)
x = torch.randn(g.N * H * W, Ci, device="cuda")
wf = torch.randn(g.K, Co, device="cuda") * 0.05
out = torch.empty(g.M, Co, device="cuda")
lib = _lib.load()
import ctypes as C
for name in ("tpgsr_halo3_trace",):
    getattr(lib, name).restype, getattr(lib, name).argtypes = C.c_int, [C.c_void_p]
with K.conv_terms(terms):
    K.make_bf_twin(wf, Ci)
    a = K.make_conv_args(g, x, wf, out)
for which, (on, tracefn) in (("whole-CU (halo3)", (1, lib.tpgsr_halo3_trace)), ("two workgroups per CU (halo)", (0, lib.tpgsr_halo_trace))):
    lib.tpgsr_halo3_set_enabled(on)
    for _ in range(3):
        K.conv_fwd(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.conv_fwd(a)
    e1.record()
    torch.cuda.synchronize()
    print(f"==== {which}: {100 * e0.elapsed_time(e1):.1f} us per launch (10 launches, untraced)")
    buf = torch.zeros(8 * 8 * 256, dtype=torch.int64, device="cuda")
    _lib.check(tracefn(buf.data_ptr()), "trace on")
    K.conv_fwd(a)
    torch.cuda.synchronize()
    _lib.check(tracefn(None), "trace off")
    t = buf.cpu().view(8, 8, 256)
    t0 = int(t[t > 0].min())
    us = lambda v: (int(v) - t0) / 100.0 if v > 0 else float("nan")
    for wg in (0, 1):
        for wave, role in ((4, "producer"), (0, "consumer")):
            row = t[wg, wave]
            nitem = int((row[:252].view(63, 4)[:, 0] > 0).sum())
            print(f"-- workgroup {wg} wave {wave} ({role}), {nitem} items, kernel entry {us(row[255]):.2f}; us since first stamp")
            for j in range(min(nitem, 10)):
                s = row[4 * j: 4 * j + 4]
                if role == "producer":
                    print(f"   item {j:2d}: loads issued {us(s[0]):7.2f}  stored {us(s[1]):7.2f}  past barrier {us(s[2]):7.2f}")
                else:
                    print(f"   item {j:2d}: at barrier {us(s[0]):7.2f}  past {us(s[1]):7.2f}  mfma issued {us(s[2]):7.2f}  tile stored {us(s[3]):7.2f}")
    if which.startswith("whole"):
        row = t[0, 0]
        taps = [us(row[100 + k]) for k in range(40) if row[100 + k] > 0]
        print("   consumer wave 0, end of tap k (us):", " ".join(f"{v:.2f}" for v in taps))
        print("   per tap (ns):", " ".join(f"{1e3 * (b - a_):.0f}" for a_, b in zip(taps[:-1], taps[1:])))
    last = max(int(v) for v in t.flatten() if v > 0)
    print(f"last stamp of the 8 traced workgroups: {(last - t0) / 100.0:.2f} us")
lib.tpgsr_halo3_set_enabled(1)
