"""`loss.gradient_loss.GradientPriorLoss` (reference: loss/gradient_loss.py:10-28).  The reference copy of this class
is broken (`@staticmethod def gradient_map(self, x)`) and never instantiated; the name is kept importable and bound to
the working image_loss implementation."""
from .image_loss import GradientPriorLoss  # noqa: F401
