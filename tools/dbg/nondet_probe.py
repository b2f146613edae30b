"""which forward kernel is not repeatable?  (round 5 debugging)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import tpgsr_oracle as O
from tpgsr_amd import kernels as K
from tpgsr_amd.model import tsrn
DEV = "cuda"
lr, hr = O.synthetic_batch(48, 4242)
lr = lr.to(DEV)
sd = O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True), 909, tps_hw=(16, 64))
def build():
    net = tsrn.TSRN(STN=True, mask=True); net.load_state_dict(sd); return net.to(DEV)
for mode in ("eval", "train"):
    outs = []
    for rep in range(3):
        net = build()
        net.train(mode == "train")
        with torch.no_grad():
            y = net(lr).clone()
        ws = next(iter(net._engine()._plans.values()))["ws"].t
        outs.append((y, {k: v.clone() for k, v in ws.items() if isinstance(k, str) and isinstance(v, torch.Tensor) and v.dtype == torch.float32}))
    torch.cuda.synchronize()
    print(mode, "outputs equal:", torch.equal(outs[0][0], outs[1][0]), torch.equal(outs[0][0], outs[2][0]))
    bad = [k for k in outs[0][1] if k in outs[1][1] and not torch.equal(outs[0][1][k], outs[1][1][k])]
    print("  workspace tensors that differ between two identical runs:", sorted(bad)[:40])
    if mode == "eval":
        perm = torch.randperm(48, generator=torch.Generator().manual_seed(1)).to(DEV)
        net = build().eval()
        with torch.no_grad():
            y = net(lr).clone(); ws1 = {k: v.clone() for k, v in next(iter(net._engine()._plans.values()))["ws"].t.items() if isinstance(k, str) and v.dtype == torch.float32}
            yp = net(lr[perm].contiguous()).clone(); ws2 = {k: v.clone() for k, v in next(iter(net._engine()._plans.values()))["ws"].t.items() if isinstance(k, str) and v.dtype == torch.float32}
        print("  permutation equivariance:", torch.equal(yp, y[perm]))
        for k in ("c1", "b1", "r0_y1", "r0_a1", "r0_y2", "r0_h1", "r0_out", "r4_out", "y7", "ups", "mups", "Pt"):
            if k in ws1:
                a, b = ws1[k], ws2[k]
                n = a.shape[0] // 48
                print("   ", k, torch.equal(b.view(48, n, -1), a.view(48, n, -1)[perm]))
