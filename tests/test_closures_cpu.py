"""CPU: small contract closures around the hot path --
* the reference's checkpoint dict (interfaces/base.py:546-585: {'state_dict_G': netG.module.state_dict(), 'info': ..., 'param_num': ...}
  + one state_dict file per recogniser) saved and loaded back through tpgsr_amd.distributed.DataParallel's `.module`;
* LmdbDatasetReal (dataset/dataset.py:104-149) against an in-memory environment with the reference's key layout, including the
  skip-undecodable-record fallback (:141-146);
* API shapes of utils/ssim_psnr.create_window (:23-27) and an empty batch through ResizeNormalize."""
import io
import os

import numpy as np
import pytest
import torch


class _Txn:
    def __init__(self, d):
        self.d = d

    def get(self, k):
        return self.d.get(k)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class FakeLmdbEnv:
    """the two calls LmdbDatasetReal makes on an lmdb.Environment: begin(write=False) -> txn with .get(key)"""

    def __init__(self, records):
        self.d = records

    def begin(self, write=False):
        assert write is False
        return _Txn(self.d)


def _png(arr):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(arr).save(b, format="PNG")
    return b.getvalue()


def test_lmdb_dataset_real_on_a_fake_environment():
    from tpgsr_amd.data import LmdbDatasetReal, lmdb_keys
    rng = np.random.default_rng(0)
    recs, truth = {b"num-samples": b"4"}, []
    for i in range(4):
        hr = rng.integers(0, 256, (32 + i, 128 + 2 * i, 3), dtype=np.uint8)
        lr = rng.integers(0, 256, (16 + i, 64 + i, 3), dtype=np.uint8)
        k = lmdb_keys(i)
        recs[k["label"]] = f"Word-{i}!".encode()
        recs[k["image_hr"]], recs[k["image_lr"]] = _png(hr), _png(lr)
        truth.append((hr, lr))
    assert lmdb_keys(0)["image_hr"] == b"image_hr-000000001" and lmdb_keys(0)["label"] == b"label-000000001"   # 1-based, 9 digits
    recs[lmdb_keys(1)["image_hr"]] = b"this is not an image"           # record 1 cannot be decoded
    ds = LmdbDatasetReal(env=FakeLmdbEnv(recs), voc_type="lower")
    assert len(ds) == 4
    hr0, lr0, s0 = ds[0]
    assert np.array_equal(hr0, truth[0][0]) and np.array_equal(lr0, truth[0][1]) and s0 == "word0"       # str_filt('lower')
    hr1, lr1, s1 = ds[1]                                               # falls through to the next record, like the reference
    assert np.array_equal(hr1, truth[2][0]) and s1 == "word2"
    assert LmdbDatasetReal(env=FakeLmdbEnv(recs), voc_type="all")[3][2] == "Word-3!"


def test_checkpoint_dict_round_trip_through_dataparallel_module(tmp_path):
    from tpgsr_amd.distributed import DataParallel
    from tpgsr_amd.model import tsrn
    from tpgsr_amd.model.crnn import crnn
    from tpgsr_amd.utils.synthetic import init_by_recipe
    net = init_by_recipe(tsrn.TSRN_TL(scale_factor=2, width=128, height=32, STN=True, srb_nums=5, mask=True, hidden_units=32), 5)
    stu = init_by_recipe(crnn.CRNN(32, 1, 37, 256), 6)
    netG, rec = DataParallel(net, broadcast=False), DataParallel(stu, broadcast=False)
    save_dict = {                                                           # interfaces/base.py:556-565
        "state_dict_G": netG.module.state_dict(),
        "info": {"arch": "tsrn_tl_cascade", "iters": 7, "epochs": 1, "batch_size": 48, "voc_type": "all", "up_scale_factor": 2},
        "best_history_res": {"easy": 0.0}, "best_model_info": {},
        "param_num": sum(p.nelement() for p in netG.module.parameters()),
        "converge": [],
    }
    torch.save(save_dict, os.path.join(tmp_path, "checkpoint.pth"))
    torch.save(rec.module.state_dict(), os.path.join(tmp_path, "recognizer_0.pth"))     # base.py:577-581
    assert save_dict["param_num"] == 3545869                                  # TSRN_TL parameter count
    # resume (interfaces/base.py:295-328: model.load_state_dict(torch.load(resume)['state_dict_G'])) into fresh wrappers
    net2 = DataParallel(tsrn.TSRN_TL(scale_factor=2, width=128, height=32, STN=True, srb_nums=5, mask=True, hidden_units=32), broadcast=False)
    stu2 = DataParallel(crnn.CRNN(32, 1, 37, 256), broadcast=False)
    ck = torch.load(os.path.join(tmp_path, "checkpoint.pth"))
    net2.module.load_state_dict(ck["state_dict_G"])
    stu2.module.load_state_dict(torch.load(os.path.join(tmp_path, "recognizer_0.pth")))
    for a, b in ((net, net2.module), (stu, stu2.module)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        assert all(torch.equal(sa[k], sb[k]) for k in sa)
    # the wrapper itself saves under the 'module.' prefix, which the reference's loaders strip (base.py:601-603)
    assert all(k.startswith("module.") for k in netG.state_dict())
    assert ck["info"]["arch"] == "tsrn_tl_cascade" and ck["info"]["iters"] == 7


def test_create_window_shape_and_empty_batch():
    from tpgsr_amd.utils.ssim_psnr import SSIM, create_window
    w = create_window(11, 3)
    assert tuple(w.shape) == (3, 1, 11, 11) and torch.equal(w[0], w[2]) and abs(float(w[0].sum()) - 1.0) < 1e-6
    assert tuple(SSIM().window.shape) == (11, 11)
    os.environ.setdefault("TPGSR_PLAN_DRYRUN", "0")
    from tpgsr_amd import kernels as K
    from tpgsr_amd.data import ResizeNormalize
    if K.DRYRUN:
        out = ResizeNormalize((128, 32), mask=True, device="cpu")([])
        assert tuple(out.shape) == (0, 4, 32, 128)
    else:
        with pytest.raises(RuntimeError):
            ResizeNormalize((128, 32), mask=True, device="cpu")([])     # GPU only, also for an empty batch


def test_bench_raises_the_hardware_queue_count_before_hip_initialises():
    """the step's three streams + RCCL's need more than HIP's default four hardware queues: bench.py sets
    GPU_MAX_HW_QUEUES=8 before it imports torch, unless the caller chose a value"""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = "import os, sys; sys.path.insert(0, %r); import bench; print(os.environ['GPU_MAX_HW_QUEUES'])" % root
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300).stdout.strip() == "8"
    env["GPU_MAX_HW_QUEUES"] = "6"
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300).stdout.strip() == "6"
