"""ResNet18 / ResNet50 (the reference's local fallback classes, ResNet.py:7-198) on the MI355X
engine.  Quirks kept on purpose because they define the numerics: ``bn1`` is applied twice (after
the stem conv and again after the max-pool, ResNet.py:130,134), the global pool is an
AdaptiveMaxPool2d stored under the name ``avgpool`` (ResNet.py:112), bottleneck stride sits on
the first 1x1 (ResNet.py:57)."""
import torch.nn as nn

from mi355.engine import Net


class BasicBlock(nn.Module):
    def __init__(self, input_channel, output_channel, stride=1, padding=1):
        super().__init__()
        self.input_channel, self.output_channel = input_channel, output_channel
        self.conv1 = nn.Conv2d(input_channel, output_channel, (3, 3), stride=stride, padding=1, bias=False)
        self.conv2 = nn.Conv2d(output_channel, output_channel, (3, 3), stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(output_channel)
        self.bn2 = nn.BatchNorm2d(output_channel)
        self.relu = nn.ReLU(inplace=True)
        self.identity = nn.Sequential()
        if stride != 1 or input_channel != output_channel:
            self.identity = nn.Sequential(nn.Conv2d(input_channel, output_channel, (1, 1), stride=stride, padding=0, bias=False),
                                          nn.BatchNorm2d(output_channel))

    def lower(self, g, x):
        idn = g.conv_bn_act(x, self.identity[0], self.identity[1], act=False) if len(self.identity) else x
        y = g.conv_bn_act(x, self.conv1, self.bn1, act=True)
        return g.conv_bn_act(y, self.conv2, self.bn2, act=True, res=idn)      # relu(bn2(conv2) + identity)


class BottleNeckBlock(nn.Module):
    def __init__(self, input_channel, output_channel, stride=1, padding=1):
        super().__init__()
        self.input_channel, self.output_channel = input_channel, output_channel
        mid = output_channel // 4
        self.conv1 = nn.Conv2d(input_channel, mid, (1, 1), stride=stride, padding=0, bias=False)
        self.conv2 = nn.Conv2d(mid, mid, (3, 3), stride=1, padding=1, bias=False)
        self.conv3 = nn.Conv2d(mid, output_channel, (1, 1), stride=1, padding=0, bias=False)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm2d(mid), nn.BatchNorm2d(mid), nn.BatchNorm2d(output_channel)
        self.relu = nn.ReLU(inplace=True)
        self.identity = nn.Sequential()
        if stride != 1 or input_channel != output_channel:
            self.identity = nn.Sequential(nn.Conv2d(input_channel, output_channel, (1, 1), stride=stride, padding=0, bias=False),
                                          nn.BatchNorm2d(output_channel))

    def lower(self, g, x):
        idn = g.conv_bn_act(x, self.identity[0], self.identity[1], act=False) if len(self.identity) else x
        y = g.conv_bn_act(x, self.conv1, self.bn1, act=True)
        y = g.conv_bn_act(y, self.conv2, self.bn2, act=True)
        return g.conv_bn_act(y, self.conv3, self.bn3, act=True, res=idn)


class _ResNet(Net):
    BLOCK, PLAN = None, ()

    def __init__(self, num_classes=3):
        super().__init__()
        self.num_classes = num_classes
        self.input_channel = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU()
        self.maxpool = nn.MaxPool2d(stride=2, kernel_size=(3, 3), padding=1)
        for i, (width, count, stride) in enumerate(self.PLAN, start=1):
            setattr(self, f"layer{i}", self.make_layer(self.BLOCK, width, count, stride))
        self.avgpool = nn.AdaptiveMaxPool2d((1, 1))
        self.flatten = nn.Flatten()
        self.fc = nn.Linear(self.PLAN[-1][0], self.num_classes)

    def make_layer(self, block, out_channel, numblocks, stride):
        layers = []
        for s in [stride] + [1] * (numblocks - 1):
            layers.append(block(self.input_channel, out_channel, s))
            self.input_channel = out_channel
        return nn.Sequential(*layers)

    def build(self, g, x):
        t = g.conv_bn_act(x, self.conv1, self.bn1, act=True)
        t = g.maxpool(t, 3, 2, 1)
        t = g.bn_act(t, self.bn1, act=False)                 # second application of the same bn1
        for i in range(1, len(self.PLAN) + 1):
            for blk in getattr(self, f"layer{i}"):
                t = blk.lower(g, t)
        v = g.global_pool(t, is_max=True)
        g.head(self.fc, v)


class ResNet18(_ResNet):
    BLOCK, PLAN = BasicBlock, ((64, 2, 1), (128, 2, 2), (256, 2, 2), (512, 2, 2))


class ResNet50(_ResNet):
    BLOCK, PLAN = BottleNeckBlock, ((256, 3, 1), (512, 4, 2), (1024, 6, 2), (2048, 3, 2))
