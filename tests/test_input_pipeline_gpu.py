"""GPU: the batched resize + normalise + mask kernels (csrc/preprocess.hip) BIT-EXACT against the oracle (itself pinned
bit-for-bit against Pillow) and against Pillow's committed outputs, on a batch of variable-size images."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import input_pipeline as ip  # noqa: E402


def test_resize_normalize_batch_bit_exact(golden_dir):
    from tpgsr_amd.data import AlignCollate, ResizeNormalize
    g = np.load(os.path.join(golden_dir, "next_resize.npz"))
    imgs = [g[f"img{j}"] for j in range(6)]
    for tag, size in (("hr", (128, 32)), ("lr", (64, 16))):
        out = ResizeNormalize(size, mask=True)(imgs).cpu().numpy()
        assert out.shape == (6, 4, size[1], size[0])
        for j in range(6):
            assert np.array_equal(out[j, :3], np.transpose(g[f"{tag}{j}"].astype(np.float32) / 255.0, (2, 0, 1))), (tag, j)
            assert np.array_equal(out[j, 3], g[f"{tag}{j}_mask"].astype(np.float32) / 255.0), (tag, j)
    # a larger ragged batch against the oracle, without the mask channel too
    rng = np.random.default_rng(3)
    batch = [rng.integers(0, 256, (int(rng.integers(6, 70)), int(rng.integers(8, 260)), 3), dtype=np.uint8) for _ in range(48)]
    for mask in (True, False):
        out = ResizeNormalize((128, 32), mask=mask)(batch).cpu().numpy()
        for j, im in enumerate(batch):
            assert np.array_equal(out[j], ip.resize_normalize(im, (128, 32), mask=mask)), j
    hr, lr, labels = AlignCollate()(list(zip(batch[:8], batch[8:16], [f"w{i}" for i in range(8)])))
    assert hr.shape == (8, 4, 32, 128) and lr.shape == (8, 4, 16, 64) and labels[3] == "w3"
    assert np.array_equal(lr[2].cpu().numpy(), ip.resize_normalize(batch[10], (64, 16), mask=True))
