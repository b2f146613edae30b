"""MORAN (reference: model/moran/moran.py:6-22)."""
from torch import nn

from .asrn_res import ASRN
from .morn import MORN


class MORAN(nn.Module):
    def __init__(self, nc, nclass, nh, targetH, targetW, BidirDecoder=False, inputDataType="torch.cuda.FloatTensor", maxBatch=256, CUDA=True):
        super().__init__()
        self.MORN = MORN(nc, targetH, targetW, inputDataType, maxBatch, CUDA)
        self.ASRN = ASRN(targetH, nc, nclass, nh, BidirDecoder, CUDA)

    def forward(self, x, length, text, text_rev, test=False, debug=False):
        """x (N, nc, H, W) NCHW; returns what the reference returns in test mode: the class scores of the first length[b] decode steps
        of every sample, concatenated ((sum(length), nclass); a pair (left-to-right, right-to-left) with BidirDecoder).  debug=True
        returns (preds, None): the reference's second element is a matplotlib / cv2 visualisation of the offsets (morn.py:81-137),
        which no caller consumes (interfaces/super_resolution.py:1391-1393 reads element 0 only)."""
        x_rectified = self.MORN(x, test, debug=False)
        preds = self.ASRN(x_rectified, length, text, text_rev, test)
        return (preds, None) if debug else preds
