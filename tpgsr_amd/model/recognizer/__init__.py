"""ASTER evaluation recognizer (reference: model/recognizer/*): STN head + TPS rectification -> ResNet_ASTER encoder with a two-layer
BiLSTM -> attention GRU decoder, greedy decode.  Same constructors and state_dict layout as the reference, so `aster_demo.pth.tar`
loads unchanged; forward runs on HIP kernels only.  Evaluation only (the reference never trains it: interfaces/base.py:831-842)."""
from .recognizer_builder import RecognizerBuilder  # noqa: F401
from .attention_recognition_head import AttentionRecognitionHead  # noqa: F401
from .resnet_aster import ResNet_ASTER  # noqa: F401
