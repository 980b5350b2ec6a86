"""VGG16 / VGG19 without BatchNorm + 3-layer MLP head (reference VGG.py:3-152) on the MI355X
engine; ``features.N`` / ``classifier.N`` indices match the reference Sequentials.

``VGG16_BN`` / ``VGG19_BN`` are the torchvision ``vgg16_bn`` / ``vgg19_bn`` layouts the reference asks
``torch.hub`` for first (helpers.py:158-166, pipeline.py:82-89: ``name + "_bn"``): [Conv3x3 -> BatchNorm ->
ReLU] stacks, ``avgpool = AdaptiveAvgPool2d((7, 7))``, ``classifier = Linear(25088, 4096) -> ReLU ->
Dropout -> Linear(4096, 4096) -> ReLU -> Dropout -> Linear(4096, num_classes)`` — same ``state_dict``
keys and shapes, so torchvision checkpoints load."""
import torch.nn as nn

from mi355.engine import Net

_CFG16 = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
_CFG19 = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")


def _features(cfg):
    layers, cin = [], 3
    for c in cfg:
        if c == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, c, kernel_size=3, stride=1, padding=1), nn.ReLU(inplace=True)]
            cin = c
    return nn.Sequential(*layers)


class VGG16_Features(nn.Module):
    def __init__(self):
        super().__init__()
        self.features = _features(_CFG16)


class VGG19_Features(nn.Module):
    def __init__(self):
        super().__init__()
        self.features = _features(_CFG19)


class Classifier(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.classifier = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(512, 256), nn.ReLU(inplace=True),
                                        nn.Dropout(0.3), nn.Linear(256, 256), nn.ReLU(inplace=True), nn.Dropout(0.3),
                                        nn.Linear(256, num_classes))


class _VGG(Net):
    FEATURES = None

    def __init__(self, num_classes):
        super().__init__()
        self.features = self.FEATURES().features
        self.classifier = Classifier(num_classes).classifier

    def build(self, g, x):
        t = g.seq(self.features, x)
        g.head(self.classifier, t)


class VGG16(_VGG):
    FEATURES = VGG16_Features


class VGG19(_VGG):
    FEATURES = VGG19_Features


def _features_bn(cfg):
    layers, cin = [], 3
    for c in cfg:
        if c == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, c, kernel_size=3, padding=1), nn.BatchNorm2d(c), nn.ReLU(inplace=True)]
            cin = c
    return nn.Sequential(*layers)


class _VGG_BN(Net):
    CFG = None

    def __init__(self, num_classes=1000, dropout=0.5):
        super().__init__()
        self.features = _features_bn(self.CFG)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(p=dropout),
                                        nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(p=dropout),
                                        nn.Linear(4096, num_classes))

    def build(self, g, x):
        t = g.seq(self.features, x)
        v = g.seq([self.avgpool], t)
        g.head(self.classifier, v)


class VGG16_BN(_VGG_BN):
    CFG = _CFG16


class VGG19_BN(_VGG_BN):
    CFG = _CFG19
