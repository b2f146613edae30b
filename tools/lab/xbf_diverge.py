#!/usr/bin/env python
"""Where do the f32 and x3 arithmetic modes diverge inside a network?  Runs the same CRNN (and TSRN_TL) step in both modes and
compares every workspace tensor, in creation order."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import tpgsr_oracle as O  # noqa: E402
from tpgsr_amd import kernels as K  # noqa: E402
from tpgsr_amd.model.crnn import crnn  # noqa: E402

DEV = "cuda"


def run(mode, sd, gray, gl):
    K.set_conv_prec(mode)
    net = crnn.CRNN(32, 1, 37, 256)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    y = net(gray.to(DEV))
    (y * gl.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    K.set_conv_prec("f32")
    eng = net._engine()
    ws = list(eng._plans.values())[0]["ws"].t
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    return {k: v.clone() for k, v in ws.items() if isinstance(v, torch.Tensor)}, grads, net


def main():
    sd = O.recipe_state_dict(O.crnn_spec(), 19)
    lr, _ = O.synthetic_batch(3, 8)
    gray = O.parse_crnn_data(lr)
    gl = torch.randn(26, 3, 37, generator=torch.Generator().manual_seed(2))
    wa, ga, _ = run("f32", sd, gray, gl)
    wb, gb, _ = run("x3", sd, gray, gl)
    wc, gc, _ = run("f32", sd, gray, gl)
    print("workspace tensors (creation order): rel L2 diff x3 vs f32 | f32 vs f32 (run-to-run)")
    for k in wa:
        if k in wb and wa[k].shape == wb[k].shape and wa[k].dtype == torch.float32:
            den = wa[k].double().norm().item() or 1.0
            d1 = (wa[k].double() - wb[k].double()).norm().item() / den
            d2 = (wa[k].double() - wc[k].double()).norm().item() / den
            flag = " <<<" if d1 > 1e-4 else ""
            print(f"  {str(k):28s} {tuple(wa[k].shape)!s:22s} {d1:.3e} | {d2:.3e}{flag}")
    print("parameter gradients:")
    for n in ga:
        den = ga[n].double().norm().item() or 1.0
        d1 = (ga[n].double() - gb[n].double()).norm().item() / den
        print(f"  {n:40s} {d1:.3e}{' <<<' if d1 > 1e-4 else ''}")


if __name__ == "__main__":
    main()
