#!/usr/bin/env python
"""Round 6: the HOST time of the drop-in loop (bench.py's `module_api` leg = the reference's loop body, interfaces/super_resolution.py:
295-424, on the nn.Module API), statement by statement: time.perf_counter() stamps behind every statement, no synchronisation inside the
step (the GPU runs behind), averaged over the timed steps.  Un-inflated counterpart of module_api_profile.py's cProfile run.
usage: python tools/lab/module_api_host.py [steps] [fused]      (fused: tpgsr_amd.optim.FusedAdam in place of torch.optim.Adam + clip)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tpgsr_amd import kernels as K  # noqa: E402
from tpgsr_amd.interfaces.super_resolution import parse_crnn_data  # noqa: E402
from tpgsr_amd.loss.image_loss import ImageLoss  # noqa: E402
from tpgsr_amd.loss.semantic_loss import SemanticLoss  # noqa: E402
from tpgsr_amd.model import tsrn  # noqa: E402
from tpgsr_amd.model.crnn import crnn  # noqa: E402
from tpgsr_amd.utils.synthetic import init_by_recipe  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    fused = len(sys.argv) > 2 and sys.argv[2] == "fused"
    dev = torch.device("cuda", 0)
    K.set_conv_prec("x2")
    images_lr, images_hr = bench.synthetic_batch(48, 1234, dev)
    model = init_by_recipe(tsrn.TSRN_TL(scale_factor=2, width=128, height=32, STN=True, srb_nums=5, mask=True, hidden_units=32), 11).to(dev).train()
    aster = init_by_recipe(crnn.CRNN(32, 1, 37, 256), 12).to(dev).eval()
    stu_model = init_by_recipe(crnn.CRNN(32, 1, 37, 256), 13).to(dev).train()
    for q in aster.parameters():
        q.requires_grad = False
    image_crit, sem_loss = ImageLoss(gradient=True, loss_weight=[1, 1e-4]), SemanticLoss()
    if fused:
        from tpgsr_amd.optim import FusedAdam
        optimizer_G = FusedAdam([model, stu_model], lr=1e-3, betas=(0.5, 0.999), clip_modules=[model], max_norm=0.25)
    else:
        optimizer_G = torch.optim.Adam(list(model.parameters()) + list(stu_model.parameters()), lr=1e-3, betas=(0.5, 0.999))
    drop_vec = torch.ones(48).float()
    drop_vec[:12] = 0.
    drop_vec = drop_vec.to(dev).view(-1, 1, 1, 1)
    acc = {}
    order = []

    def stamp(name, t0):
        t1 = time.perf_counter()
        if name not in acc:
            acc[name] = 0.0
            order.append(name)
        acc[name] += t1 - t0
        return t1

    def loop_body(rec):
        t = time.perf_counter()
        g_hr = parse_crnn_data(images_hr[:, :3, :, :]);                         t = stamp("parse_crnn_data(hr)", t) if rec else t
        t_logits = aster(g_hr);                                                   t = stamp("aster(...) teacher forward", t) if rec else t
        label_vecs_hr = torch.nn.functional.softmax(t_logits.detach(), -1);       t = stamp("softmax(teacher)", t) if rec else t
        g_lr = parse_crnn_data(images_lr[:, :3, :, :]);                         t = stamp("parse_crnn_data(lr)", t) if rec else t
        label_vecs_logits = stu_model(g_lr);                                      t = stamp("stu_model(...) student forward", t) if rec else t
        label_vecs = torch.nn.functional.softmax(label_vecs_logits, -1)
        label_vecs_final = label_vecs.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2);   t = stamp("softmax + permutes (student)", t) if rec else t
        loss_recog_distill = sem_loss(label_vecs, label_vecs_hr) * 100;           t = stamp("sem_loss * 100", t) if rec else t
        label_vecs_final = label_vecs_final * drop_vec;                           t = stamp("prior * drop_vec", t) if rec else t
        cascade_images = model(images_lr, label_vecs_final);                      t = stamp("model(images_lr, prior) SR forward", t) if rec else t
        loss_img = image_crit(cascade_images, images_hr).mean() * 100;            t = stamp("image_crit(...).mean() * 100", t) if rec else t
        loss_im = loss_img + loss_recog_distill;                                  t = stamp("loss sum", t) if rec else t
        optimizer_G.zero_grad();                                                  t = stamp("optimizer.zero_grad()", t) if rec else t
        loss_im.backward();                                                       t = stamp("loss.backward()", t) if rec else t
        if not fused:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25);             t = stamp("clip_grad_norm_", t) if rec else t
        optimizer_G.step();                                                       t = stamp("optimizer.step()", t) if rec else t
        return loss_im

    for _ in range(8):
        loop_body(False)
    torch.cuda.synchronize()
    # inside the module calls: the plan executor (tpgsr_plan_run3: every launch + stream edge of a recorded plan, in C++) by plan, and the
    # engines' backward entry points (called from autograd's thread)
    plan_t, plan_n = {}, {}
    orig_run = K.Plan.run

    def timed_run(self):
        t0 = time.perf_counter()
        orig_run(self)
        key = f"{self.name} ({len(self.ops)} ops)"
        plan_t[key] = plan_t.get(key, 0.0) + time.perf_counter() - t0
        plan_n[key] = plan_n.get(key, 0) + 1
    K.Plan.run = timed_run
    bwd_t = {}
    for tag, mod in (("TSRN_TL", model), ("CRNN student", stu_model)):
        eng = mod._engine()
        ob = eng.backward

        def wrapped(*a, _ob=ob, _tag=tag, **k):
            t0 = time.perf_counter()
            r = _ob(*a, **k)
            bwd_t[_tag] = bwd_t.get(_tag, 0.0) + time.perf_counter() - t0
            return r
        eng.backward = wrapped
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = loop_body(True)
    t_sub = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"optimizer: {'tpgsr_amd.optim.FusedAdam' if fused else 'torch.optim.Adam + clip_grad_norm_'}")
    print(f"step {1e3 * dt / steps:.3f} ms, host submission {1e3 * t_sub / steps:.3f} ms per step, loss {float(loss.item()):.5f}")
    for name in order:
        print(f"  {name:44s} {1e3 * acc[name] / steps:7.3f} ms")
    print(f"  {'(sum of the stamps)':44s} {1e3 * sum(acc.values()) / steps:7.3f} ms")
    print("inside loss.backward(): the engines' backward entry points (autograd's thread)")
    for tag, v in bwd_t.items():
        print(f"  {tag + ' engine.backward':44s} {1e3 * v / steps:7.3f} ms")
    print("plan executor (C++), by plan:")
    tot = 0.0
    for key in sorted(plan_t, key=lambda k: -plan_t[k]):
        tot += plan_t[key]
        print(f"  {key:44s} {1e3 * plan_t[key] / steps:7.3f} ms   ({plan_n[key] / steps:.1f} runs per step)")
    print(f"  {'(all plans)':44s} {1e3 * tot / steps:7.3f} ms")


if __name__ == "__main__":
    main()
