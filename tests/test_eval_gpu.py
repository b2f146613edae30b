"""GPU: the evaluation path: on-device CTC greedy decoding, PSNR / SSIM reduction kernels
against fixtures computed by the reference's own utils/metrics.py / utils/ssim_psnr.py (tests/golden/next_eval_metrics.npz),
and the multi-stage inference loop (TextSREvaluator) against the oracle's restatement of TextSR.eval."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tpgsr_oracle as O  # noqa: E402

DEV = "cuda"


def test_ctc_psnr_ssim_kernels_vs_reference_fixture(golden_dir):
    from tpgsr_amd.utils import metrics, ssim_psnr
    g = np.load(os.path.join(golden_dir, "next_eval_metrics.npz"))
    logits = torch.tensor(g["logits"]).to(DEV)
    got = metrics.get_string_crnn(logits)
    assert got == [str(s) for s in g["strings"]]
    labels, lengths = metrics.ctc_greedy(logits)
    assert lengths.cpu().tolist() == [len(s) for s in got] and int(labels.min()) >= -1
    a, b = torch.tensor(g["a"]).to(DEV), torch.tensor(g["b"]).to(DEV)
    p = float(ssim_psnr.calculate_psnr(a, b))
    s = float(ssim_psnr.SSIM()(a, b))
    print(f"psnr {p:.6f} (reference {float(g['psnr']):.6f}), ssim {s:.7f} (reference {float(g['ssim']):.7f})")
    assert abs(p - float(g["psnr"])) < 1e-4 and abs(s - float(g["ssim"])) < 2e-6
    assert abs(float(ssim_psnr.calculate_psnr(a[:, :3].contiguous(), b[:, :3].contiguous())) - float(g["psnr3"])) < 1e-4
    assert float(ssim_psnr.calculate_psnr(a, a)) == float("inf")
    assert abs(float(ssim_psnr.SSIM()(a, a)) - float(g["ssim_same"])) < 2e-6
    assert [metrics.str_filt(str(t), "lower") for t in g["filt_in"]] == [str(t) for t in g["filt_lower"]]


@pytest.mark.parametrize("stu_iter", [1, 3])
def test_multi_stage_inference_loop_vs_oracle(stu_iter):
    from tpgsr_amd.interfaces.super_resolution import TextSREvaluator
    from tpgsr_amd.model import tsrn
    from tpgsr_amd.model.crnn import crnn
    sd_sr = O.recipe_state_dict(O.tsrn_spec(STN=True, mask=True, text_prior=True), 41, tps_hw=(16, 64))
    sr = tsrn.TSRN_TL(STN=True, mask=True)
    sr.load_state_dict(sd_sr)
    sd_t, tpgs = [], []
    for k in range(stu_iter):
        sd = O.recipe_state_dict(O.crnn_spec(), 50 + k)
        m = crnn.CRNN(32, 1, 37, 256)
        m.load_state_dict(sd)
        tpgs.append(m.to(DEV).eval())
        sd_t.append(sd)
    sd_r = O.recipe_state_dict(O.crnn_spec(), 60)
    rec = crnn.CRNN(32, 1, 37, 256)
    rec.load_state_dict(sd_r)
    lr, hr = O.synthetic_batch(6, 99)
    ev = TextSREvaluator([sr.to(DEV).eval()], tpgs, rec.to(DEV).eval(), stu_iter=stu_iter, sr_share=True, tpg_share=False)
    labels = ["abc", "7x", "", "hello", "q", "zz9"]
    out = ev.eval_batch(lr.to(DEV), hr.to(DEV), labels)
    torch.cuda.synchronize()
    ref = O.tpgsr_eval_step([O.as_params(sd_sr, False)], [O.as_params(x, False) for x in sd_t], O.as_params(sd_r, False), lr, hr,
                            stu_iter=stu_iter, sr_share=True, tpg_share=False)
    for i in range(stu_iter):
        e = (out["images_sr"][i].cpu() - ref["sr"][i]).abs().max().item()
        print(f"stage {i}: SR max err {e:.2e}")
        assert e < 2e-5 * (3 ** i)
        assert torch.equal(out["priors"][i].cpu().permute(1, 0, 2).argmax(-1), ref["priors"][i].argmax(-1))
    assert abs(float(out["psnr"]) - float(ref["psnr"])) < 1e-3                # the north_star gate, evaluation path
    assert abs(float(out["ssim"]) - float(ref["ssim"])) < 1e-5
    assert out["pred_sr"] == ref["pred_sr"] and out["pred_lr"] == ref["pred_lr"] and out["pred_hr"] == ref["pred_hr"]
    assert out["n_correct_sr"] == sum(a == b for a, b in zip(ref["pred_sr"], labels))
    with pytest.raises(RuntimeError):
        sr.train()
        ev.super_resolve(lr.to(DEV))
