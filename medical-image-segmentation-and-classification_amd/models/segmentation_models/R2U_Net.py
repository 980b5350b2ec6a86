"""R2U-Net (recurrent-residual U-Net) on the MI355X engine — drop-in for
models/segmentation_models/R2U_Net.py:50-111 (t=5 default comes from the net, not the block)."""
import torch.nn as nn

from mi355.engine import Net
from ._blocks import WIDTHS, Recurrent_block, RRCNN_block, UpConv  # noqa: F401  (re-exported names)


class _R2Base(Net):
    GATED = False

    def __init__(self, in_channels=3, out_channels=1, t=5):
        super().__init__()
        self.max_pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.upsample = nn.Upsample(scale_factor=2)
        cin = in_channels
        for i, w in enumerate(WIDTHS, start=1):
            setattr(self, f"RRCNN{i}", RRCNN_block(cin, w, t=t))
            cin = w
        for lvl in (5, 4, 3, 2):
            w = WIDTHS[lvl - 2]
            setattr(self, f"up{lvl}", UpConv(2 * w, w))
            if self.GATED:
                from ._blocks import AttentionGate
                setattr(self, f"att{lvl}", AttentionGate(F_g=w, F_l=w, F_int=w // 2))
            setattr(self, f"up_RRCNN{lvl}", RRCNN_block(2 * w, w, t=t))
        self.conv_1x1 = nn.Conv2d(64, out_channels, kernel_size=1, stride=1, padding=0)
        self._out_channels = out_channels

    def build(self, g, x):
        skips, t = {}, x
        for i in range(1, 6):
            if i > 1:
                t = g.maxpool(t, 2, 2, 0)
            t = getattr(self, f"RRCNN{i}").lower(g, t)
            skips[i] = t
        d = skips[5]
        for lvl in (5, 4, 3, 2):
            skip = skips[lvl - 1]
            cat, (lo, hi) = g.new_cat(skip.N, skip.H, skip.W, [skip.C, skip.C])
            d = g.seq(getattr(self, f"up{lvl}").up, d, out=hi)
            if self.GATED:
                g.gate(getattr(self, f"att{lvl}"), g=d, x=skip, out=lo)
            else:
                g.copy(skip, lo)
            d = getattr(self, f"up_RRCNN{lvl}").lower(g, cat)
        g.logit_conv(d, self.conv_1x1)


class R2U_Net(_R2Base):
    GATED = False
