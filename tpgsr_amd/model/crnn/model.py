"""`--tpg OPT`: the None-ResNet-None-CTC recogniser of the reference as text-prior generator (reference
model/crnn/model.py:25-95 with main.py:60-75's option set: no transformation, ResNet features, no sequence model, CTC head).
gray (N, 1, 32, 100) -> logits (T = 26, N, num_class), seq-first like `crnn.CRNN`, so `TextSR.train` uses either unchanged.
Same constructor (an `opt` object with the reference's attribute names) and state_dict keys."""
from torch import nn

from ... import functional as Fh
from ..nn_params import EngineHolder, LinearParams, _NoForward
from .modules.feature_extraction import ResNet_FeatureExtractor


class Model(EngineHolder, nn.Module):
    def __init__(self, opt):
        super().__init__()
        g = (lambda k, d=None: opt.get(k, d)) if isinstance(opt, dict) else (lambda k, d=None: getattr(opt, k, d))
        self.opt = opt
        self.stages = {"Trans": g("Transformation"), "Feat": g("FeatureExtraction"), "Seq": g("SequenceModeling"), "Pred": g("Prediction")}
        if self.stages != {"Trans": "None", "Feat": "ResNet", "Seq": "None", "Pred": "CTC"}:
            raise NotImplementedError(f"the TPGSR path selects None-ResNet-None-CTC (main.py:60-75), got {self.stages}")
        self.FeatureExtraction = ResNet_FeatureExtractor(g("input_channel"), g("output_channel"))
        self.FeatureExtraction_output = g("output_channel")
        self.AdaptiveAvgPool = _NoForward()          # nn.AdaptiveAvgPool2d((None, 1)) = mean over the height axis
        self.SequenceModeling_output = self.FeatureExtraction_output
        self.Prediction = LinearParams(self.SequenceModeling_output, g("num_class"))

    def _engine(self):
        """engine adapter (tpgsr_amd/engine_functional.py): lets TPGSRTrainStep / FusedAdam / ArenaPool drive this recogniser as a
        student or teacher of the fused train step (`--tpg OPT`, interfaces/super_resolution.py:77-80)"""
        eng = self.__dict__.get("_eng")
        if eng is None:
            from ...engine_functional import FunctionalEngine
            eng = FunctionalEngine(self)
            self.__dict__["_eng"] = eng
        return eng

    def forward(self, input, text=None, is_train=True):
        feat = self.FeatureExtraction(Fh.to_nhwc(input))          # (N, h, 26, 512) NHWC
        seq = Fh.mean_over_height(feat)                           # (N, 26, 512) == the reference's permute + pool + squeeze
        pred = self.Prediction(seq)                               # (N, 26, num_class)
        return pred.permute(1, 0, 2)
