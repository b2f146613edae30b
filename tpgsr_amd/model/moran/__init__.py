"""MORAN evaluation recognizer (reference: model/moran/*), `--test_model MORAN`: MORN rectifier + ASRN attention recognizer.
Evaluation only (the reference loads it frozen, interfaces/base.py:587-606); same state_dict keys / shapes."""
from .moran import MORAN  # noqa: F401
