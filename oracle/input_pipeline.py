"""CPU restatement (numpy, integer arithmetic) of the reference's input pipeline contract -- TEST INFRASTRUCTURE, like the
rest of oracle/: only tests/ and bench/smoke checkers import it; the product path is tpgsr_amd/data.py + csrc/preprocess.hip.

reference: dataset/dataset.py:615-632 `resizeNormalize(size, mask, interpolation=Image.BICUBIC)`:
    img.resize(size, BICUBIC) -> ToTensor (uint8 HWC -> float CHW / 255) -> [mask: img.convert('L'); thres = mean; 255 where
    L <= thres else 0; ToTensor; cat as 4th channel]
and its callers alignCollate_real* (dataset/dataset.py:1226-1323: HR -> (imgW, imgH) = (128, 32), LR -> (64, 16)).
`img.resize` is Pillow's two-pass (horizontal, then vertical) separable resampling on 8-bit data (third-party: Pillow,
src/libImaging/Resample.c -- not vendored in the reference; restated from its published algorithm and PINNED bit-for-bit
against the Pillow installed in the build container by tests/test_input_pipeline_cpu.py): Keys bicubic (a = -0.5), support
2 * max(scale, 1) (antialiased when shrinking), coefficients normalised, rounded to 22-bit fixed point, accumulation from
2^21, >> 22, clamped to 0..255 after EACH pass.  `convert('L')` is (19595 R + 38470 G + 7471 B + 32768) >> 16."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int):
    """per output index: (xmin, [int32 coefficients])"""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    res = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        if ww != 0.0:
            w = [v / ww for v in w]
        k = [int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in w]
        res.append((xmin, np.array(k, dtype=np.int64)))
    return res


def _pass(img, out_size, axis):
    """img uint8 [H][W][C]; resample along axis (1 = horizontal, 0 = vertical)"""
    in_size = img.shape[axis]
    if in_size == out_size:
        return img
    src = img.astype(np.int64)
    shape = list(img.shape)
    shape[axis] = out_size
    out = np.empty(shape, dtype=np.uint8)
    for xx, (xmin, k) in enumerate(resample_coeffs(in_size, out_size)):
        if axis == 1:
            acc = (src[:, xmin:xmin + len(k), :] * k[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        else:
            acc = (src[xmin:xmin + len(k), :, :] * k[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[xx, :, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def pil_resize_bicubic(img: np.ndarray, size_wh):
    """Image.fromarray(img).resize(size_wh, Image.BICUBIC) for uint8 [H][W][3]"""
    w, h = size_wh
    return _pass(_pass(img, w, 1), h, 0)


def luma(img: np.ndarray) -> np.ndarray:
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def resize_normalize(img: np.ndarray, size_wh, mask=True) -> np.ndarray:
    """dataset/dataset.py:615-632: uint8 [H][W][3] -> float32 [3 or 4][h][w]"""
    r = pil_resize_bicubic(img, size_wh)
    t = np.transpose(r.astype(np.float32) / 255.0, (2, 0, 1))
    if not mask:
        return t
    L = luma(r)
    thres = L.mean()
    m = np.where(L > thres, 0, 255).astype(np.float32) / 255.0
    return np.concatenate([t, m[None]], 0)


# LMDB key contract of the reference's datasets (dataset/dataset.py:104-149): 1-based, 9-digit
def lmdb_keys(index: int):
    i = index + 1
    return dict(label=b"label-%09d" % i, image_hr=b"image_hr-%09d" % i, image_lr=b"image_lr-%09d" % i)


NUM_SAMPLES_KEY = b"num-samples"
