"""R2AttU-Net (recurrent-residual blocks + attention gates) on the MI355X engine — drop-in for
models/segmentation_models/R2AttU_Net.py:88-158."""
from ._blocks import AttentionGate, Recurrent_block, RRCNN_block, UpConv  # noqa: F401
from .R2U_Net import _R2Base


class R2AttU_Net(_R2Base):
    GATED = True
