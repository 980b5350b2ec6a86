"""Attention U-Net on the MI355X launch-plan engine — drop-in for the reference class
(models/segmentation_models/AttentionUNet.py:56-121): same constructor, attribute names,
``state_dict`` layout and ``forward(x: NCHW float) -> logits [N,1,H,W]``."""
from mi355.engine import Net
from ._blocks import WIDTHS, AttentionGate, UpConv, conv_bn_relu_x2
import torch.nn as nn

basic_block = conv_bn_relu_x2


class AttentionUNet(Net):
    def __init__(self, in_channel=3, out_channel=1):
        super().__init__()
        self.in_channel, self.out_channel = in_channel, out_channel
        self.max_pool = nn.MaxPool2d(kernel_size=2, stride=2)
        cin = 3            # the reference hard-codes basic_block(3, 64) and only stores in_channel (AttentionUNet.py:59,62)
        for i, w in enumerate(WIDTHS, start=1):                       # conv1..conv5
            setattr(self, f"conv{i}", conv_bn_relu_x2(cin, w))
            cin = w
        for lvl in (5, 4, 3, 2):                                       # up5/att5/up_conv5 ... registered level by level
            w = WIDTHS[lvl - 2]
            setattr(self, f"up{lvl}", UpConv(2 * w, w))
            setattr(self, f"att{lvl}", AttentionGate(F_g=w, F_l=w, F_int=w // 2))
            setattr(self, f"up_conv{lvl}", conv_bn_relu_x2(2 * w, w))
        self.out = nn.Conv2d(64, out_channel, kernel_size=1, stride=1, padding=0)

    def build(self, g, x):
        skips = {}
        t = x
        for i in range(1, 6):
            if i > 1:
                t = g.maxpool(t, 2, 2, 0)
            t = g.seq(getattr(self, f"conv{i}"), t)
            skips[i] = t
        d = skips[5]
        for lvl in (5, 4, 3, 2):
            skip = skips[lvl - 1]
            cat, (lo, hi) = g.new_cat(skip.N, skip.H, skip.W, [skip.C, skip.C])    # torch.cat((x_att, d), 1)
            d = g.seq(getattr(self, f"up{lvl}").up, d, out=hi)
            g.gate(getattr(self, f"att{lvl}"), g=d, x=skip, out=lo)
            d = g.seq(getattr(self, f"up_conv{lvl}"), cat)
        g.logit_conv(d, self.out)
