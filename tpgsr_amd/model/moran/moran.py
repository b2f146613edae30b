"""MORAN = MORN rectifier + ASRN recognizer behind one call (reference: model/moran/moran.py:6-22).  Constructor arguments and
state_dict prefixes (`MORN.*`, `ASRN.*`) are the reference's, so its checkpoints load; evaluation only."""
from torch import nn

from . import asrn_res, morn


class MORAN(nn.Module):
    def __init__(self, nc, nclass, nh, targetH, targetW, BidirDecoder=False, inputDataType="torch.cuda.FloatTensor", maxBatch=256, CUDA=True):
        super().__init__()
        self.config = dict(nc=nc, nclass=nclass, nh=nh, target_hw=(targetH, targetW), bidirectional_decoder=bool(BidirDecoder))
        self.add_module("MORN", morn.MORN(nc, targetH, targetW, inputDataType=inputDataType, maxBatch=maxBatch, CUDA=CUDA))
        self.add_module("ASRN", asrn_res.ASRN(imgH=targetH, nc=nc, nclass=nclass, nh=nh, BidirDecoder=BidirDecoder, CUDA=CUDA))

    def forward(self, x, length, text, text_rev, test=False, debug=False):
        """x (N, nc, H, W) NCHW.  Test mode returns what the reference returns: the class scores of the first length[b] decode steps
        of every sample, concatenated -- (sum(length), nclass), or the pair (left-to-right, right-to-left) with BidirDecoder.
        debug=True yields (preds, None): the reference's second element is a matplotlib / cv2 picture of the offsets
        (morn.py:81-137) that no caller consumes (interfaces/super_resolution.py:1391-1393 reads element 0 only)."""
        scores = self.ASRN(self.MORN(x, test), length, text, text_rev, test)
        if debug:
            return scores, None
        return scores
