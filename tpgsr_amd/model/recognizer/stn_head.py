"""STN localisation head of the recognizer (reference: model/recognizer/stn_head.py:27-95): the TSRN head's architecture with a 2x2
max-pool after each of the first five conv stages (input 3 x 32 x 64 -> 1 x 2 x 256)."""
from ..stn_head import STNHead as _STNHead


class STNHead(_STNHead):
    POOLS = [(2, 2), (2, 2), (2, 2), (2, 2), (2, 2), None]

    def __init__(self, in_planes, num_ctrlpoints, activation="none"):
        super().__init__(in_planes, num_ctrlpoints, activation)
