"""MORN, the multi-object rectification network (reference: model/moran/morn.py:6-79), test-mode path: a small CNN predicts a map of
vertical offsets, read out on the regular sampling grid (grid_sample) and added to its y coordinates; the image is re-sampled with the
displaced grid; one enhancement pass repeats this on the rectified image.  Runs on the HIP kernels one operator at a time
(tpgsr_amd/functional.py).  The training-time branch (random skip, no enhancement) is not part of the evaluation path."""
import numpy as np
import torch
from torch import nn

from ... import functional as Fh
from ..nn_params import BatchNormParams, Conv2dParams, _NoForward


class MORN(nn.Module):
    def __init__(self, nc, targetH, targetW, inputDataType="torch.cuda.FloatTensor", maxBatch=256, CUDA=True):
        super().__init__()
        self.targetH, self.targetW, self.inputDataType, self.maxBatch, self.cuda_flag = targetH, targetW, inputDataType, maxBatch, CUDA
        hold = _NoForward          # parameter-less slots keep the reference's nn.Sequential indices (state_dict keys cnn.1, cnn.2, cnn.5 ...)
        self.cnn = nn.Sequential(
            hold(), Conv2dParams(nc, 64, 3, padding=1), BatchNormParams(64), hold(), hold(),
            Conv2dParams(64, 128, 3, padding=1), BatchNormParams(128), hold(), hold(),
            Conv2dParams(128, 64, 3, padding=1), BatchNormParams(64), hold(),
            Conv2dParams(64, 16, 3, padding=1), BatchNormParams(16), hold(),
            Conv2dParams(16, 1, 3, padding=1), BatchNormParams(1))
        # the regular sampling grid (morn.py:27-43), (x, y) in [-1, 1]; one image's worth -- the reference tiles it maxBatch times
        h_list = np.arange(targetH) * 2.0 / (targetH - 1) - 1
        w_list = np.arange(targetW) * 2.0 / (targetW - 1) - 1
        g = np.stack([np.tile(w_list[None, :], (targetH, 1)), np.tile(h_list[:, None], (1, targetW))], -1)
        self._grid = torch.from_numpy(g).float()           # not a buffer: the reference keeps it out of the state_dict too
        self._grid_dev = {}

    def _base_grid(self, N, device):
        key = (N, str(device))
        if key not in self._grid_dev:
            self._grid_dev = {key: self._grid.to(device).unsqueeze(0).expand(N, -1, -1, -1).contiguous()}
        return self._grid_dev[key]

    def _offsets(self, x, grid):
        """x NHWC (N, 32, 100, nc) -> vertical offsets on the regular grid, (N, 32, 100, 1)"""
        c = self.cnn
        h = Fh.max_pool2d(x, 2, 2)
        h = Fh.max_pool2d(c[2](c[1](h), act="relu"), 2, 2)
        h = Fh.max_pool2d(c[6](c[5](h), act="relu"), 2, 2)
        h = c[10](c[9](h), act="relu")
        h = c[13](c[12](h), act="relu")
        off = c[16](c[15](h))
        pooled = Fh.signed_relu_pool_diff(off, 2, 1)
        return Fh.grid_sample(pooled, grid, (self.targetH, self.targetW), align_corners=False)

    def forward(self, x, test, enhance=1, debug=False):
        if self.training or not test:
            raise RuntimeError("MORN is an evaluation module here (interfaces/base.py:603-605 loads MORAN frozen): call .eval() and pass test=True")
        if debug:
            raise RuntimeError("the debug visualisation of the reference (matplotlib / colour / cv2, morn.py:81-137) is not provided")
        if x.shape[0] > self.maxBatch:
            raise ValueError(f"batch {x.shape[0]} > maxBatch {self.maxBatch}")
        if tuple(x.shape[2:]) != (self.targetH, self.targetW):
            raise NotImplementedError(f"MORN takes {self.targetH}x{self.targetW} inputs here (parse_moran_data always produces them); "
                                      f"got {tuple(x.shape[2:])}")
        with torch.no_grad():
            N = x.shape[0]
            xh = Fh.to_nhwc(x)                      # == the reference's bilinear resize to (targetH, targetW): identity at this size
            grid = self._base_grid(N, x.device)
            HW = (self.targetH, self.targetW)
            og = self._offsets(xh, grid)
            rect = Fh.grid_sample(xh, Fh.offset_grid_y(grid, og), HW, align_corners=False)
            for _ in range(enhance):
                og = Fh.add(og, self._offsets(rect, grid))
                rect = Fh.grid_sample(xh, Fh.offset_grid_y(grid, og), HW, align_corners=False)
            return Fh.to_nchw(rect)
