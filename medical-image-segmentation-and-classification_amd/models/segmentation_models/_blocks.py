"""Parameter containers shared by the U-Net family.

The containers are plain ``torch.nn`` modules arranged so that ``state_dict()`` keys and tensor
shapes are identical to the reference's blocks (cited per factory); they are never *called* —
``mi355.engine.Net.build`` lowers them to HIP launch plans."""
import torch.nn as nn

WIDTHS = (64, 128, 256, 512, 1024)


def conv_bn_relu_x2(cin, cout):
    """Keys 0,1,3,4 (conv, bn, conv, bn) as in AttentionUNet.py:4-13 / ResnetUnet.py:5-14."""
    layers = []
    for a, b in ((cin, cout), (cout, cout)):
        layers += [nn.Conv2d(a, b, 3, padding=1), nn.BatchNorm2d(b), nn.ReLU(inplace=True)]
    return nn.Sequential(*layers)


class UpConv(nn.Module):
    """``up.{1,2}`` = conv3x3 + BN after a nearest x2 up-sampling (AttentionUNet.py:15-27)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.up = nn.Sequential(nn.Upsample(scale_factor=2), nn.Conv2d(in_channels, out_channels, 3, 1, 1, bias=True),
                                nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))


class AttentionGate(nn.Module):
    """``W_g``, ``W_x`` (1x1 conv + BN) and ``psi`` (1x1 conv -> 1 channel + BN + sigmoid),
    AttentionUNet.py:29-54."""

    def __init__(self, F_g, F_l, F_int):
        super().__init__()
        self.W_g = nn.Sequential(nn.Conv2d(F_g, F_int, 1), nn.BatchNorm2d(F_int))
        self.W_x = nn.Sequential(nn.Conv2d(F_l, F_int, 1), nn.BatchNorm2d(F_int))
        self.psi = nn.Sequential(nn.Conv2d(F_int, 1, 1), nn.BatchNorm2d(1), nn.Sigmoid())
        self.relu = nn.ReLU(inplace=True)


class Recurrent_block(nn.Module):
    """One shared conv3x3+BN+ReLU applied t+1 times (R2AttU_Net.py:29-45)."""

    def __init__(self, in_channels, out_channels, t=2):
        super().__init__()
        self.t = t
        self.out_channels = out_channels
        self.conv = nn.Sequential(nn.Conv2d(out_channels, out_channels, 3, 1, 1, bias=True),
                                  nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))

    def lower(self, g, x, plus=None, out=None):
        """x1 = f(x); t times x1 = f(x + x1).  The weight pack is shared; BN statistics, running-stat
        updates and gradients are per application, in application order.  ``plus``: a tensor added to the block's OUTPUT in
        the last application's BatchNorm apply pass (the residual sum of RRCNN_block, R2AttU_Net.py:59: no pass of its own)."""
        conv, bn = self.conv[0], self.conv[1]
        if self.t == 0:
            return g.conv_bn_act(x, conv, bn, act=True, post_add=plus, out=out)
        # every application but the last emits s = x + relu(bn(conv(.))) directly (the next application's
        # input); x1 itself is never materialised except as the block output
        s = g.conv_bn_act(x, conv, bn, act=True, post_add=x)
        for _ in range(self.t - 1):
            s = g.conv_bn_act(s, conv, bn, act=True, post_add=x)
        return g.conv_bn_act(s, conv, bn, act=True, post_add=plus, out=out)


class RRCNN_block(nn.Module):
    """conv_1x1 -> two recurrent blocks -> residual sum (R2AttU_Net.py:47-59)."""

    def __init__(self, in_channels, out_channels, t=2):
        super().__init__()
        self.RCNN = nn.Sequential(Recurrent_block(in_channels, out_channels, t=t),
                                  Recurrent_block(in_channels, out_channels, t=t))
        self.conv_1x1 = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def lower(self, g, x, out=None):
        x0 = g.conv_act(x, self.conv_1x1, relu=False)
        # x0 + RCNN(x0): the sum rides in the last recurrent application's BatchNorm apply pass, its gradient in that pass's backward
        if not g.fuse_residual:
            return g.add(x0, self.RCNN[1].lower(g, self.RCNN[0].lower(g, x0)), out=out)
        return self.RCNN[1].lower(g, self.RCNN[0].lower(g, x0), plus=x0, out=out)
