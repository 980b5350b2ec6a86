"""ResNet-50-encoder U-Net on the MI355X engine — drop-in for the reference class
(models/segmentation_models/ResnetUnet.py:17-83): ``ResNetUnet(n_classes=1, freeze=True)``,
encoder1..5 / maxpool / decoder5..1 / out attributes, same ``state_dict`` keys.

The reference takes its encoder from ``torchvision.models.resnet50(weights=DEFAULT)`` (line 32: a
network download).  Neither torchvision nor the weights exist offline, so the encoder is restated
here from torchvision's public ResNet-50 v1.5 layout (stride on the 3x3 of each bottleneck,
``downsample.{0,1}`` shortcuts) with default random initialisation; a real torchvision
``state_dict`` loads into it unchanged.  Frozen encoder parameters keep ``requires_grad=False``
(their wgrad/dgrad launches are pruned from the plan) while their BatchNorms still run in train
mode, exactly like the reference under ``model.train()`` (helpers.py:315)."""
import torch.nn as nn

from mi355.engine import Net
from ._blocks import conv_bn_relu_x2
from models.classification_models.TorchvisionResNet import Bottleneck, make_layer   # torchvision ResNet-50 v1.5 containers

basic_block = conv_bn_relu_x2


def _layer(inplanes, planes, blocks, stride):
    return make_layer(Bottleneck, inplanes, planes, blocks, stride)


class DecoderBlock(nn.Module):
    """ConvTranspose2d(k2,s2) on the deeper feature, cat([up, skip]), double conv (ResnetUnet.py:17-27)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.basic_block = conv_bn_relu_x2(in_channels, out_channels)
        self.up_sample = nn.ConvTranspose2d(in_channels - out_channels, in_channels - out_channels, 2, 2)

    def lower(self, g, down, skip):
        cu = self.up_sample.out_channels
        cat, (lo, hi) = g.new_cat(down.N, 2 * down.H, 2 * down.W, [cu, skip.C])
        g.conv_transpose(down, self.up_sample, out=lo)
        g.copy(skip, hi)
        return g.seq(self.basic_block, cat)


class ResNetUnet(Net):
    def __init__(self, n_classes=1, freeze=True):
        super().__init__()
        self.encoder1 = nn.Sequential(nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.encoder2 = _layer(64, 64, 3, 1)
        self.encoder3 = _layer(256, 128, 4, 2)
        self.encoder4 = _layer(512, 256, 6, 2)
        self.encoder5 = _layer(1024, 512, 3, 2)
        if freeze:
            self._freeze_backbone()
        self.decoder5 = DecoderBlock(2048 + 1024, 1024)
        self.decoder4 = DecoderBlock(1024 + 512, 512)
        self.decoder3 = DecoderBlock(512 + 256, 256)
        self.decoder2 = DecoderBlock(256 + 64, 64)
        self.decoder1 = nn.Sequential(nn.ConvTranspose2d(64, 32, kernel_size=2, stride=2), nn.Conv2d(32, 32, kernel_size=3, padding=1),
                                      nn.BatchNorm2d(32), nn.ReLU(inplace=True))
        self.out = nn.Conv2d(32, n_classes, kernel_size=1)
        self._n_classes = n_classes

    def _freeze_backbone(self):
        for layer in (self.encoder1, self.encoder2, self.encoder3, self.encoder4, self.encoder5):
            for p in layer.parameters():
                p.requires_grad = False

    def build(self, g, x):
        e1 = g.seq(self.encoder1, x)
        t = g.maxpool(e1, 3, 2, 1)
        feats = []
        for enc in (self.encoder2, self.encoder3, self.encoder4, self.encoder5):
            for blk in enc:
                t = blk.lower(g, t)
            feats.append(t)
        e2, e3, e4, e5 = feats
        d = self.decoder5.lower(g, e5, e4)
        d = self.decoder4.lower(g, d, e3)
        d = self.decoder3.lower(g, d, e2)
        d = self.decoder2.lower(g, d, e1)
        d = g.seq(self.decoder1, d)
        g.logit_conv(d, self.out)
