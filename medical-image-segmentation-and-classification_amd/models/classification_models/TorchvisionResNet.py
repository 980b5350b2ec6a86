"""torchvision-layout ``resnet18`` / ``resnet50`` on the MI355X engine — the models the reference actually trains when it
is online: ``get_class_model`` asks ``torch.hub`` for ``pytorch/vision:v0.10.0`` ``resnet18`` / ``resnet50`` first
(helpers.py:158-161) and only falls back to its local ``ResNet.py`` classes when that fails (:170-185).

torchvision is not vendored by the reference and is absent offline, so the layout is restated from torchvision's public
architecture (ResNet v1.5): ``conv1`` 7x7/2 (no bias) - ``bn1`` - ReLU - MaxPool 3x3/2 - ``layer1..4`` - global AVERAGE pool -
``fc``; BasicBlock = [conv3x3(stride) - BN - ReLU - conv3x3 - BN] + shortcut, Bottleneck = [1x1 - BN - ReLU - 3x3(stride) - BN -
ReLU - 1x1 - BN] + shortcut, shortcut = ``downsample.{0,1}`` (1x1 conv stride s + BN) where the shape changes.  Same
``state_dict`` keys / shapes as torchvision (a hub checkpoint loads unchanged; 11 689 512 / 25 557 032 parameters at 1000
classes).  Parity unpinned at this boundary (no torchvision here to generate vectors): checked against the oracle's
restatement, which composes primitives pinned by the reference's own blocks."""
import torch.nn as nn

from mi355.engine import Net


class BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock parameter container."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def lower(self, g, x):
        idn = x if self.downsample is None else g.conv_bn_act(x, self.downsample[0], self.downsample[1], act=False)
        y = g.conv_bn_act(x, self.conv1, self.bn1, act=True)
        return g.conv_bn_act(y, self.conv2, self.bn2, act=True, res=idn)


class Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (v1.5: the stride sits on the 3x3) parameter container."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def lower(self, g, x):
        idn = x if self.downsample is None else g.conv_bn_act(x, self.downsample[0], self.downsample[1], act=False)
        y = g.conv_bn_act(x, self.conv1, self.bn1, act=True)
        y = g.conv_bn_act(y, self.conv2, self.bn2, act=True)
        return g.conv_bn_act(y, self.conv3, self.bn3, act=True, res=idn)


def make_layer(block, inplanes, planes, blocks, stride):
    """torchvision ``ResNet._make_layer``: a ``downsample`` shortcut exactly where stride != 1 or the width changes."""
    ds = None
    if stride != 1 or inplanes != planes * block.expansion:
        ds = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                           nn.BatchNorm2d(planes * block.expansion))
    mods = [block(inplanes, planes, stride, ds)]
    mods += [block(planes * block.expansion, planes) for _ in range(blocks - 1)]
    return nn.Sequential(*mods)


class ResNet(Net):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers), start=1):
            setattr(self, f"layer{i}", make_layer(block, inplanes, planes, n, 1 if i == 1 else 2))
            inplanes = planes * block.expansion
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)

    def build(self, g, x):
        t = g.conv_bn_act(x, self.conv1, self.bn1, act=True)
        t = g.maxpool(t, 3, 2, 1)
        for i in range(1, 5):
            for blk in getattr(self, f"layer{i}"):
                t = blk.lower(g, t)
        v = g.global_pool(t, is_max=False)
        g.head(self.fc, v)


def resnet18(num_classes=1000):
    return ResNet(BasicBlock, (2, 2, 2, 2), num_classes)


def resnet50(num_classes=1000):
    return ResNet(Bottleneck, (3, 4, 6, 3), num_classes)
