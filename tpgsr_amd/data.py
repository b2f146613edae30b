"""Input pipeline contract of the reference on the device.

reference: dataset/dataset.py:615-632 `resizeNormalize(size, mask, interpolation=Image.BICUBIC)` -- Pillow bicubic
`img.resize`, `ToTensor`, luminance-threshold mask channel -- applied per image by `alignCollate_real*`
(dataset/dataset.py:1226-1323: HR -> (128, 32), LR -> (64, 16)) on ONE DataLoader worker; LMDB record keys
dataset/dataset.py:104-149.  Here a whole batch of variable-size uint8 HWC images is resized + normalised + masked by three
kernel launches (csrc/preprocess.hip), bit-exact against Pillow's 8-bit resampling; the host only concatenates the decoded
pixels and Pillow's per-size coefficient tables (cached per (in, out) size pair).  JPEG/PNG decoding and LMDB I/O stay host
work exactly as in the reference (`buf2PIL`)."""
import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import _lib, kernels as K

NUM_SAMPLES_KEY = b"num-samples"


def lmdb_keys(index: int) -> Dict[str, bytes]:
    """record keys of sample `index` (0-based) in the reference's LMDB layout (1-based, 9 digits)"""
    i = index + 1
    return dict(label=b"label-%09d" % i, image_hr=b"image_hr-%09d" % i, image_lr=b"image_lr-%09d" % i)


class ResizeNormalize:
    """`resizeNormalize((w, h), mask)` for a batch: list of uint8 [H][W][3] arrays -> float tensor (N, 3 + mask, h, w) on `device`."""

    def __init__(self, size: Tuple[int, int], mask: bool = False, device="cuda"):
        self.size, self.mask, self.device = (int(size[0]), int(size[1])), bool(mask), torch.device(device)
        self._tables: Dict[Tuple[int, int], Tuple[np.ndarray, np.ndarray, int]] = {}

    def _coeffs(self, in_size, out_size):
        key = (in_size, out_size)
        t = self._tables.get(key)
        if t is None:
            lib = _lib.load()
            ks = lib.tpgsr_resample_ksize(in_size, out_size)
            b = np.zeros((out_size, 2), np.int32)
            k = np.zeros((out_size, ks), np.int32)
            _lib.check(lib.tpgsr_resample_coeffs(in_size, out_size, b.ctypes.data, k.ctypes.data), "tpgsr_resample_coeffs")
            t = self._tables[key] = (b, k, ks)
        return t

    def __call__(self, images: Sequence[np.ndarray]) -> torch.Tensor:
        if not K.DRYRUN and self.device.type != "cuda":
            raise RuntimeError("tpgsr_amd.data.ResizeNormalize runs on the GPU only (the reference's PIL path is the CPU form)")
        ow, oh = self.size
        N = len(images)
        if N == 0:                                        # an empty batch is an empty tensor, not a numpy concatenate error
            return torch.empty(0, 4 if self.mask else 3, oh, ow, dtype=torch.float32, device=self.device)
        descs = (_lib.ImageDesc * N)()
        tabs: List[np.ndarray] = []
        toff, poff, maxH = 0, 0, 1
        tab_index: Dict[Tuple[int, int], Tuple[int, int, int]] = {}

        def table(in_size, out_size):
            nonlocal toff
            key = (in_size, out_size)
            if key not in tab_index:
                b, k, ks = self._coeffs(in_size, out_size)
                tab_index[key] = (toff, toff + b.size, ks)
                tabs.extend([b.reshape(-1), k.reshape(-1)])
                toff += b.size + k.size
            return tab_index[key]

        flat = []
        for d, img in zip(descs, images):
            a = np.ascontiguousarray(img)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise ValueError(f"expected uint8 [H][W][3] images, got {a.dtype} {a.shape}")
            H, W = a.shape[:2]
            d.offset, d.H, d.W = poff, H, W
            d.xb_off, d.xk_off, d.kx = table(W, ow)
            d.yb_off, d.yk_off, d.ky = table(H, oh)
            flat.append(a.reshape(-1))
            poff += a.size
            maxH = max(maxH, H)
        dev = self.device
        pix = torch.from_numpy(np.concatenate(flat)).to(dev)
        tab = torch.from_numpy(np.concatenate(tabs)).to(dev)
        dsc = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        Cc = 4 if self.mask else 3
        tmp = torch.empty(N * maxH * ow * 3, dtype=torch.uint8, device=dev)
        res8 = torch.empty(N * oh * ow * 3, dtype=torch.uint8, device=dev)
        out = torch.empty(N, Cc, oh, ow, dtype=torch.float32, device=dev)
        K._launch("tpgsr_resize_normalize", K._p(pix), K._p(dsc), K._p(tab), N, oh, ow, maxH, int(self.mask), K._p(tmp), K._p(res8), K._p(out))
        return out


class AlignCollate:
    """`alignCollate_real*` (dataset/dataset.py:1226-1323) reduced to what the training loop consumes: a list of
    (HR image, LR image, label string) -> (images_HR (N, C, imgH, imgW), images_lr (N, C, imgH/ds, imgW/ds), label_strs);
    with `labels=True` (`alignCollate_realWTLAMask`, what `--use_label` trains on) also label_vecs (N, 37, 1, L) one-hot,
    weighted_mask (all label indices concatenated) and weighted_tics (N) -- the three tensors TPGSRTrainStep.step(labels=) takes."""

    D2A = "-0123456789abcdefghijklmnopqrstuvwxyz"      # dataset/dataset.py:1108-1116: index 0 is the CTC blank

    def __init__(self, imgH=32, imgW=128, down_sample_scale=2, mask=True, device="cuda", labels=False):
        self.hr = ResizeNormalize((imgW, imgH), mask, device)
        self.lr = ResizeNormalize((imgW // down_sample_scale, imgH // down_sample_scale), mask, device)
        self.labels = bool(labels)
        self.a2d = {ch: i for i, ch in enumerate(self.D2A)}

    def encode(self, label_strs):
        """dataset/dataset.py:1255-1323: lower-case, words of 15 characters and more cut to 15, characters outside the alphabet dropped;
        a word without any label gets one blank (index 0) and weight 0; max_len = the longest word (not label list)"""
        alsize = len(self.D2A)
        lists, max_len = [], 0
        for word in label_strs:
            word = word.lower()
            if len(word) >= 15:
                word = word[:15]
            lists.append([self.a2d[ch] for ch in word if ch in self.a2d])
            max_len = max(max_len, len(word))
        label_vecs = torch.zeros(len(lists), max(max_len, 1), alsize)
        mask, tics = [], []
        for i, ll in enumerate(lists):
            if ll:
                label_vecs[i, torch.arange(len(ll)), torch.tensor(ll)] = 1.0
                mask.extend(ll)
                tics.append(1)
            else:
                label_vecs[i, 0, 0] = 1.0
                mask.append(0)
                tics.append(0)
        return label_vecs.unsqueeze(1).permute(0, 3, 1, 2).contiguous(), torch.tensor(mask, dtype=torch.long), torch.tensor(tics)

    def __call__(self, batch):
        images_hr, images_lr, label_strs = zip(*batch)
        out = (self.hr(images_hr), self.lr(images_lr), list(label_strs))
        return out + self.encode(label_strs) if self.labels else out


class LmdbDatasetReal:
    """`lmdbDataset_real` (dataset/dataset.py:104-149): (HR, LR, label) triples from the reference's LMDB layout, decoded to
    uint8 arrays for AlignCollate.  Needs `lmdb` (unless an opened environment is handed in as `env`) and `PIL`, like the reference.
    As in the reference, a record whose images cannot be decoded is SKIPPED: `self[index]` returns the next sample (:141-146)."""

    def __init__(self, root=None, voc_type="upper", max_len=100, test=False, env=None):
        if env is None:
            try:
                import lmdb
            except ImportError as e:
                raise ImportError("LmdbDatasetReal needs the `lmdb` package (as the reference's dataset/dataset.py does)") from e
            env = lmdb.open(root, max_readers=1, readonly=True, lock=False, readahead=False, meminit=False)
        if not env:
            raise RuntimeError(f"cannot open lmdb from {root}")
        self.env = env
        with self.env.begin(write=False) as txn:
            self.nSamples = int(txn.get(NUM_SAMPLES_KEY))
        self.voc_type, self.max_len, self.test = voc_type, max_len, test

    def __len__(self):
        return self.nSamples

    def __getitem__(self, index):
        import io
        from PIL import Image
        from .utils.metrics import str_filt
        assert index <= len(self), "index range error"               # (the reference's own guard, dataset.py:131)
        for probe in range(index, index + len(self) + 1):              # at most one lap: the reference recurses on self[index + 1]
            keys = lmdb_keys(probe % max(1, len(self)))
            with self.env.begin(write=False) as txn:
                word, buf_hr, buf_lr = txn.get(keys["label"]), txn.get(keys["image_hr"]), txn.get(keys["image_lr"])
            try:
                if word is None or buf_hr is None or buf_lr is None:
                    raise IOError(f"missing record {probe}")
                hr = np.asarray(Image.open(io.BytesIO(buf_hr)).convert("RGB"))
                lr = np.asarray(Image.open(io.BytesIO(buf_lr)).convert("RGB"))
            except (IOError, OSError, SyntaxError, ValueError):
                continue
            return hr, lr, str_filt(word.decode(), self.voc_type)
        raise IOError("no decodable record in the LMDB")
