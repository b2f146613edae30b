"""RecognizerBuilder (reference: model/recognizer/recognizer_builder.py:27-104): the integrated ASTER recognizer.  Evaluation path with
the greedy decoder: rectification (bilinear down-sample to 32x64 -> STN head -> TPS to 32x100), encoder, AttentionRecognitionHead.sample."""
import torch
from torch import nn

from ... import functional as Fh
from ..tps_spatial_transformer import TPSSpatialTransformer
from .attention_recognition_head import AttentionRecognitionHead
from .resnet_aster import ResNet_ASTER
from .stn_head import STNHead

tps_inputsize = [32, 64]
tps_outputsize = [32, 100]
num_control_points = 20
tps_margins = [0.05, 0.05]


class RecognizerBuilder(nn.Module):
    def __init__(self, arch, rec_num_classes, sDim=512, attDim=512, max_len_labels=100, eos="EOS", STN_ON=True):
        super().__init__()
        self.arch, self.rec_num_classes, self.sDim, self.attDim = arch, rec_num_classes, sDim, attDim
        self.max_len_labels, self.eos, self.STN_ON = max_len_labels, eos, STN_ON
        self.tps_inputsize = tps_inputsize
        self.encoder = ResNet_ASTER(self.arch)
        self.decoder = AttentionRecognitionHead(num_classes=rec_num_classes, in_planes=self.encoder.out_planes, sDim=sDim, attDim=attDim,
                                                max_len_labels=max_len_labels)
        if self.STN_ON:
            self.tps = TPSSpatialTransformer(output_image_size=tuple(tps_outputsize), num_control_points=num_control_points,
                                             margins=tuple(tps_margins))
            self.stn_head = STNHead(in_planes=3, num_ctrlpoints=num_control_points, activation="none")

    def rectify(self, x):
        """(N, 3, H, W) in [-1, 1] -> (rectified (N, 3, 32, 100), control points (N, 20, 2))"""
        stn_in = Fh.to_nchw(Fh.interpolate_bilinear(Fh.to_nhwc(x), tuple(self.tps_inputsize)))
        _, ctrl = self.stn_head(stn_in)
        rect, _ = self.tps(x, ctrl)
        return rect, ctrl

    def forward(self, input_dict):
        """evaluation: {'images': (N, 3, 32, 128) in [-1, 1], ...} -> {'losses': {}, 'output': {'pred_rec', 'pred_rec_score'}} with the
        GREEDY decode (the reference calls beam_search here, which raises on torch >= 1.5; its loss against dummy targets is not computed)"""
        if self.training:
            raise RuntimeError("RecognizerBuilder is an evaluation recognizer here (interfaces/base.py:831-842 loads it frozen); call .eval()")
        x = input_dict["images"] if isinstance(input_dict, dict) else input_dict
        with torch.no_grad():
            out = {"losses": {}, "output": {}}
            if self.STN_ON:
                x, ctrl = self.rectify(x)
                out["output"]["ctrl_points"], out["output"]["rectified_images"] = ctrl, x
            feats = self.encoder(x)
            out["output"]["encoder_feats"] = feats
            ids, scores = self.decoder.sample([feats, None, None])
            out["output"]["pred_rec"], out["output"]["pred_rec_score"] = ids, scores
            return out
