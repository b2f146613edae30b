"""Round 6: the persistent BiLSTM launches ALONE on the chip (bs 48, T 26, Hh 256): microseconds per launch and per time step of the
data-tagged forward (`tpgsr_lstm_seq_fwdg`), the counter forward and the backward recurrence, HIP events over REPS back-to-back launches;
the data-tagged forward must equal the counter forward bit for bit.   usage: python tools/lab/lstm_seq_time.py [N] [T] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from tpgsr_amd import kernels as K  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
T = int(sys.argv[2]) if len(sys.argv) > 2 else 26
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 300
Hh = 256
dev = torch.device("cuda")
g = torch.Generator().manual_seed(7)
Gin = torch.randn(N, T, 2, 4 * Hh, generator=g).to(dev)
WT = (torch.randn(2, Hh, 4 * Hh, generator=g) / Hh ** 0.5).to(dev)
b = torch.randn(2, 4 * Hh, generator=g).to(dev)
dout = torch.randn(N, T, 2 * Hh, generator=g).to(dev)
hx, sync = K.lstm_seq_buffers(dev)
hg, gsync = K.lstm_seq_granule_buffers(dev)
px, bsync = K.lstm_seq_bwd_buffers(dev)
w = [WT[d].t().contiguous() for d in range(2)]


def timed(fn, reps=REPS):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


res = {}
for name, fn, bufs in (("fwd (counter)", lambda G, C, o: K.lstm_seq_fwd(G, WT, b, C, o, hx, sync, N, T, Hh), None),
                       ("fwdg (data-tagged)", lambda G, C, o: K.lstm_seq_fwdg(G, WT, b, C, o, hg, gsync, N, T, Hh), None)):
    G = Gin.clone()
    Cst = torch.empty(N, T, 2, Hh, device=dev)
    out = torch.empty(N, T, 2 * Hh, device=dev)
    fn(G, Cst, out)
    torch.cuda.synchronize()
    res[name] = (G.clone(), Cst.clone(), out.clone())
    # the timed launches re-activate activated gates: numerically meaningless, same instruction stream and traffic
    us = timed(lambda: fn(G, Cst, out))
    print(f"{name:22s} {us:7.1f} us per launch  {us / T:5.2f} us per step")
same = all(torch.equal(a, c) for a, c in zip(res["fwd (counter)"], res["fwdg (data-tagged)"]))
print("data-tagged == counter forward, bitwise:", same)
G, Cst, _ = res["fwd (counter)"]
Gb = G.clone()
us = timed(lambda: K.lstm_seq_bwd(Gb, Cst, dout, w[0], w[1], px, bsync, N, T, Hh))
print(f"{'bwd (counter)':22s} {us:7.1f} us per launch  {us / T:5.2f} us per step")
assert int(sync[2].item()) == 0 and int(gsync[2].item()) == 0 and int(bsync[2].item()) == 0, "a hand-off timed out"
assert same
