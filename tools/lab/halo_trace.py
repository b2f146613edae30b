#!/usr/bin/env python
"""time line of the halo conv kernel (tpgsr_halo_trace): per-item stamps of producer wave 4 and consumer wave 0 of workgroup 0/1
usage: halo_trace.py N H W Cin Cout KH KW pad mode"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tpgsr_amd import _lib, kernels as K  # noqa: E402

N, H, W, Ci, Co, KH, KW, pad = (int(v) for v in sys.argv[1:9])
mode = sys.argv[9]
g = K.ConvGeom(N, H, W, Ci, Co, KH, KW, pad, pad)
x = torch.randn(g.N * H * W, Ci, device="cuda")
wf = torch.randn(g.K, Co, device="cuda") * 0.05
out = torch.empty(g.M, Co, device="cuda")
K.make_bf_twin(wf, Ci)
K.set_conv_prec(mode)
a = K.make_conv_args(g, x, wf, out)
for _ in range(3):
    K.conv_fwd(a)
torch.cuda.synchronize()
buf = torch.zeros(8 * 8 * 256, dtype=torch.int64, device="cuda")
lib = _lib.load()
_lib.check(lib.tpgsr_halo_trace(buf.data_ptr()), "trace on")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
K.conv_fwd(a)
e1.record()
torch.cuda.synchronize()
_lib.check(lib.tpgsr_halo_trace(None), "trace off")
print(f"kernel (traced): {1e3 * e0.elapsed_time(e1):.1f} us")
t = buf.cpu().view(8, 8, 256)
t0 = int(t[t > 0].min())
us = lambda v: (int(v) - t0) / 100.0 if v > 0 else float("nan")
for wg in (0, 1):
    for wave, role in ((4, "producer"), (0, "consumer")):
        row = t[wg, wave]
        nitem = int((row.view(64, 4)[:, 0] > 0).sum())
        print(f"-- workgroup {wg} wave {wave} ({role}), {nitem} items; us since first stamp")
        for j in range(min(nitem, 14)):
            s = row[4 * j: 4 * j + 4]
            if role == "producer":
                print(f"   item {j:2d}: loads issued {us(s[0]):7.2f}  stored {us(s[1]):7.2f}  past barrier {us(s[2]):7.2f}")
            else:
                print(f"   item {j:2d}: at barrier {us(s[0]):7.2f}  past {us(s[1]):7.2f}  mfma issued {us(s[2]):7.2f}  tile stored {us(s[3]):7.2f}")
last = max(int(v) for v in t.flatten() if v > 0)
print(f"last stamp of the 8 traced workgroups: {(last - t0) / 100.0:.2f} us")
