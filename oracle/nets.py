"""Functional CPU restatement of the reference model zoo (TEST INFRASTRUCTURE).

Every network is a pure function ``f(sd, x, training)`` over a flat
``state_dict``-style mapping ``sd`` whose keys and NCHW shapes are exactly the
reference's (so a reference checkpoint drives the oracle unchanged).  Nothing
here is imported by the product path.

Reference anchors (``/root/reference`` paths):
  * AttentionUNet ........ models/segmentation_models/AttentionUNet.py:4-121
  * R2U_Net .............. models/segmentation_models/R2U_Net.py:4-111
  * R2AttU_Net ........... models/segmentation_models/R2AttU_Net.py:29-158
  * ResNet18 / ResNet50 .. models/classification_models/ResNet.py:7-198
  * VGG16 / VGG19 ........ models/classification_models/VGG.py:3-152
  * ResNetUnet ........... models/segmentation_models/ResnetUnet.py:17-83
    (encoder = torchvision ResNet-50 v1.5 layout; parity unpinned at that
    boundary because torchvision is absent from the build container)
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------
# leaf ops
# ----------------------------------------------------------------------------
def _conv(sd, p, x, stride=1, pad=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=pad)


def _bn(sd, p, x, training):
    """BatchNorm2d, torch defaults (eps 1e-5, momentum 0.1): batch statistics and
    running-stat EMA in training, running stats in eval."""
    if training:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(
        x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
        training, BN_MOMENTUM, BN_EPS)


# ----------------------------------------------------------------------------
# kinks: the network's non-smooth decisions (ReLU masks, max-pool arg-max)
# ----------------------------------------------------------------------------
class Kinks:
    """Record / replay of every ReLU mask and max-pool arg-max of one forward, in call order.

    The gradient of these networks is discontinuous in the weights and inputs: an fp32 evaluation that lands a
    pre-activation on the other side of zero than fp64 (a dozen of ~10^7 elements do) changes that element's gradient by
    100 % and every parameter gradient by ~1e-3 relative — for ANY fp32 implementation, the reference's CPU path included.
    Replaying the masks of one evaluation (``replay``: ``relu(x) := x * mask``, ``maxpool := gather at the recorded
    arg-max``) inside another makes the two evaluate the SAME smooth function, so that their gradients can be compared to
    rounding accuracy (tests/test_gpu_kinks.py).  Off (``mode is None``) the wrappers are plain F.relu / F.max_pool2d."""
    mode = None
    relu, pool = [], []
    i_relu = i_pool = 0

    @classmethod
    def start(cls, mode, relu=None, pool=None):
        cls.mode, cls.relu, cls.pool = mode, (relu if relu is not None else []), (pool if pool is not None else [])
        cls.i_relu = cls.i_pool = 0

    @classmethod
    def stop(cls):
        r, p = cls.relu, cls.pool
        used = (cls.i_relu, cls.i_pool)
        cls.mode, cls.relu, cls.pool = None, [], []
        return r, p, used


def _relu(x):
    if Kinks.mode == "record":
        Kinks.relu.append((x > 0).detach())
    elif Kinks.mode == "replay":
        m = Kinks.relu[Kinks.i_relu]
        Kinks.i_relu += 1
        assert m.shape == x.shape, (m.shape, x.shape)
        return x * m.to(x.dtype)
    return F.relu(x)


def _maxpool(x, k, s, p=0):
    if Kinks.mode == "record":
        y, idx = F.max_pool2d(x, k, s, p, return_indices=True)
        Kinks.pool.append(idx.detach())
        return y
    if Kinks.mode == "replay":
        idx = Kinks.pool[Kinks.i_pool]
        Kinks.i_pool += 1
        return x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
    return F.max_pool2d(x, k, s, p)


def _gmaxpool(x):
    """AdaptiveMaxPool2d((1, 1)) + flatten (ResNet.py:112,140) with the arg-max as a recorded / replayed kink."""
    n, c = x.shape[:2]
    flat = x.flatten(2)
    if Kinks.mode == "replay":
        idx = Kinks.pool[Kinks.i_pool]
        Kinks.i_pool += 1
        return flat.gather(2, idx.view(n, c, 1)).view(n, c)
    v, idx = flat.max(2)
    if Kinks.mode == "record":
        Kinks.pool.append(idx.detach())
    return v


def _up2(x):
    return F.interpolate(x, scale_factor=2.0, mode="nearest")


def _linear(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


# ----------------------------------------------------------------------------
# U-Net family blocks
# ----------------------------------------------------------------------------
def double_conv(sd, p, x, tr):
    """basic_block (AttentionUNet.py:4-13): indices 0 conv,1 bn,3 conv,4 bn."""
    x = _relu(_bn(sd, p + ".1", _conv(sd, p + ".0", x, 1, 1), tr))
    return _relu(_bn(sd, p + ".4", _conv(sd, p + ".3", x, 1, 1), tr))


def up_conv(sd, p, x, tr):
    """UpConv (AttentionUNet.py:15-27): nearest x2, conv3x3, bn, relu."""
    return _relu(_bn(sd, p + ".up.2", _conv(sd, p + ".up.1", _up2(x), 1, 1), tr))


def attention_gate(sd, p, g, x, tr):
    """AttentionGate (AttentionUNet.py:29-54)."""
    g1 = _bn(sd, p + ".W_g.1", _conv(sd, p + ".W_g.0", g), tr)
    x1 = _bn(sd, p + ".W_x.1", _conv(sd, p + ".W_x.0", x), tr)
    a = _relu(g1 + x1)
    psi = torch.sigmoid(_bn(sd, p + ".psi.1", _conv(sd, p + ".psi.0", a), tr))
    return x * psi


def recurrent(sd, p, x, t, tr):
    """Recurrent_block (R2AttU_Net.py:29-45): f(x) then t times f(x + x1)."""
    def f(z):
        return _relu(_bn(sd, p + ".conv.1", _conv(sd, p + ".conv.0", z, 1, 1), tr))
    x1 = f(x)
    for _ in range(t):
        x1 = f(x + x1)
    return x1


def rrcnn(sd, p, x, t, tr):
    """RRCNN_block (R2AttU_Net.py:47-59)."""
    x = _conv(sd, p + ".conv_1x1", x)
    x1 = recurrent(sd, p + ".RCNN.1", recurrent(sd, p + ".RCNN.0", x, t, tr), t, tr)
    return x + x1


# ----------------------------------------------------------------------------
# segmentation nets
# ----------------------------------------------------------------------------
def attention_unet(sd, x, training=False):
    tr = training
    mp = lambda z: _maxpool(z, 2, 2)
    x1 = double_conv(sd, "conv1", x, tr)
    x2 = double_conv(sd, "conv2", mp(x1), tr)
    x3 = double_conv(sd, "conv3", mp(x2), tr)
    x4 = double_conv(sd, "conv4", mp(x3), tr)
    x5 = double_conv(sd, "conv5", mp(x4), tr)
    d, skips = x5, {5: x4, 4: x3, 3: x2, 2: x1}
    for lvl in (5, 4, 3, 2):
        d = up_conv(sd, f"up{lvl}", d, tr)
        s = attention_gate(sd, f"att{lvl}", d, skips[lvl], tr)
        d = double_conv(sd, f"up_conv{lvl}", torch.cat((s, d), 1), tr)
    return _conv(sd, "out", d)


def _r2_family(sd, x, training, t, gated):
    tr = training
    mp = lambda z: _maxpool(z, 2, 2)
    x1 = rrcnn(sd, "RRCNN1", x, t, tr)
    x2 = rrcnn(sd, "RRCNN2", mp(x1), t, tr)
    x3 = rrcnn(sd, "RRCNN3", mp(x2), t, tr)
    x4 = rrcnn(sd, "RRCNN4", mp(x3), t, tr)
    x5 = rrcnn(sd, "RRCNN5", mp(x4), t, tr)
    d, skips = x5, {5: x4, 4: x3, 3: x2, 2: x1}
    for lvl in (5, 4, 3, 2):
        d = up_conv(sd, f"up{lvl}", d, tr)
        s = skips[lvl]
        if gated:
            s = attention_gate(sd, f"att{lvl}", d, s, tr)
        d = rrcnn(sd, f"up_RRCNN{lvl}", torch.cat((s, d), 1), t, tr)
    return _conv(sd, "conv_1x1", d)


def r2u_net(sd, x, training=False, t=5):
    return _r2_family(sd, x, training, t, gated=False)


def r2attu_net(sd, x, training=False, t=5):
    return _r2_family(sd, x, training, t, gated=True)


# ----------------------------------------------------------------------------
# classifiers
# ----------------------------------------------------------------------------
def _basic_block(sd, p, x, stride, tr):
    """BasicBlock (ResNet.py:7-45)."""
    idn = x
    if (p + ".identity.0.weight") in sd:
        idn = _bn(sd, p + ".identity.1", _conv(sd, p + ".identity.0", x, stride, 0), tr)
    y = _relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 1), tr))
    y = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", y, 1, 1), tr)
    return _relu(y + idn)


def _bottleneck_local(sd, p, x, stride, tr):
    """BottleNeckBlock (ResNet.py:47-91) — stride sits on the first 1x1."""
    idn = x
    if (p + ".identity.0.weight") in sd:
        idn = _bn(sd, p + ".identity.1", _conv(sd, p + ".identity.0", x, stride, 0), tr)
    y = _relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 0), tr))
    y = _relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", y, 1, 1), tr))
    y = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", y, 1, 0), tr)
    return _relu(y + idn)


def _head(sd, p, x, training, drop_mask=None):
    """``fc`` is either Linear (``fc.weight``) or Sequential(Dropout, Linear)
    (``fc.1.weight``) after helpers.add_dropout_to_fc (helpers.py:124-134).
    Dropout is the identity in eval; in training an explicit 0/1-scaled mask may
    be injected (RNG streams cannot be matched across implementations)."""
    if (p + ".1.weight") in sd:
        if training and drop_mask is not None:
            x = x * drop_mask
        return _linear(sd, p + ".1", x)
    return _linear(sd, p, x)


def _resnet_local(sd, x, training, block, counts, drop_mask):
    """ResNet18/50 (ResNet.py:95-198) including the quirks: bn1 is applied twice
    (lines 130,134 / 185,189) and the global pool is AdaptiveMaxPool2d (112/167)."""
    tr = training
    x = _conv(sd, "conv1", x, 2, 3)
    x = _relu(_bn(sd, "bn1", x, tr))
    x = _maxpool(x, 3, 2, 1)
    x = _bn(sd, "bn1", x, tr)
    for li, n in enumerate(counts, start=1):
        for b in range(n):
            stride = 2 if (li > 1 and b == 0) else 1
            x = block(sd, f"layer{li}.{b}", x, stride, tr)
    x = _gmaxpool(x)
    return _head(sd, "fc", x, tr, drop_mask)


def resnet18(sd, x, training=False, drop_mask=None):
    return _resnet_local(sd, x, training, _basic_block, (2, 2, 2, 2), drop_mask)


def resnet50(sd, x, training=False, drop_mask=None):
    return _resnet_local(sd, x, training, _bottleneck_local, (3, 4, 6, 3), drop_mask)


VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
             512, 512, 512, 512, "M")


def _vgg(sd, x, cfg, training, drop_masks):
    """VGG16/19 (VGG.py:3-152): conv3x3+bias+ReLU stacks (no BN), 2x2 max-pools,
    head = AdaptiveAvgPool(1) -> Linear 512-256 -> ReLU -> Dropout(.3) -> Linear
    256-256 -> ReLU -> Dropout(.3) -> Linear (optionally preceded by the extra
    Dropout(.5) that helpers.add_dropout_to_fc inserts, helpers.py:135-143)."""
    idx = 0
    for c in cfg:
        if c == "M":
            x = _maxpool(x, 2, 2)
            idx += 1
        else:
            x = _relu(_conv(sd, f"features.{idx}", x, 1, 1))
            idx += 2
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    dm = list(drop_masks) if (training and drop_masks is not None) else []
    x = _relu(_linear(sd, "classifier.2", x))
    if dm:
        x = x * dm[0]
    x = _relu(_linear(sd, "classifier.5", x))
    if dm:
        x = x * dm[1]
    if "classifier.9.weight" in sd:          # after add_dropout_to_fc
        if len(dm) > 2:
            x = x * dm[2]
        return _linear(sd, "classifier.9", x)
    return _linear(sd, "classifier.8", x)


def _vgg_bn(sd, x, cfg, training, drop_masks):
    """torchvision vgg16_bn / vgg19_bn (the models helpers.py:158-166 / pipeline.py:82-89 request from torch.hub),
    restated from the public torchvision layout (torchvision is absent here: PARITY UNPINNED at this boundary):
    features = [Conv3x3 p1 + bias -> BN -> ReLU]* with 2x2 max-pools, avgpool = AdaptiveAvgPool2d((7,7)), classifier =
    Linear(25088,4096) ReLU Dropout Linear(4096,4096) ReLU Dropout Linear(4096,n); add_dropout_to_fc moves the last
    Linear from classifier.6 to classifier.7 behind an extra Dropout (helpers.py:135-143)."""
    idx = 0
    for c in cfg:
        if c == "M":
            x = _maxpool(x, 2, 2)
            idx += 1
        else:
            x = _relu(_bn(sd, f"features.{idx + 1}", _conv(sd, f"features.{idx}", x, 1, 1), training))
            idx += 3
    x = F.adaptive_avg_pool2d(x, (7, 7)).flatten(1)
    dm = list(drop_masks) if (training and drop_masks is not None) else []
    x = _relu(_linear(sd, "classifier.0", x))
    if dm:
        x = x * dm[0]
    x = _relu(_linear(sd, "classifier.3", x))
    if dm:
        x = x * dm[1]
    if "classifier.7.weight" in sd:          # after add_dropout_to_fc
        if len(dm) > 2:
            x = x * dm[2]
        return _linear(sd, "classifier.7", x)
    return _linear(sd, "classifier.6", x)


def vgg16_bn(sd, x, training=False, drop_masks=None):
    return _vgg_bn(sd, x, VGG16_CFG, training, drop_masks)


def vgg19_bn(sd, x, training=False, drop_masks=None):
    return _vgg_bn(sd, x, VGG19_CFG, training, drop_masks)


def vgg16(sd, x, training=False, drop_masks=None):
    return _vgg(sd, x, VGG16_CFG, training, drop_masks)


def vgg19(sd, x, training=False, drop_masks=None):
    return _vgg(sd, x, VGG19_CFG, training, drop_masks)


# ----------------------------------------------------------------------------
# ResNetUnet (torchvision ResNet-50 v1.5 encoder restated from its public layout)
# ----------------------------------------------------------------------------
def _bottleneck_tv(sd, p, x, stride, tr):
    """torchvision Bottleneck v1.5: stride on the 3x3; shortcut = downsample.{0,1}."""
    idn = x
    if (p + ".downsample.0.weight") in sd:
        idn = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride, 0), tr)
    y = _relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, 1, 0), tr))
    y = _relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", y, stride, 1), tr))
    y = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", y, 1, 0), tr)
    return _relu(y + idn)


def _decoder_block(sd, p, down, skip, tr):
    """DecoderBlock (ResnetUnet.py:17-27): ConvT(k2,s2) -> cat([x, skip]) -> basic_block."""
    x = F.conv_transpose2d(down, sd[p + ".up_sample.weight"], sd[p + ".up_sample.bias"], stride=2)
    return double_conv(sd, p + ".basic_block", torch.cat((x, skip), 1), tr)


def resnet_unet(sd, x, training=False):
    tr = training
    e1 = _relu(_bn(sd, "encoder1.1", _conv(sd, "encoder1.0", x, 2, 3), tr))
    y = _maxpool(e1, 3, 2, 1)
    feats = []
    for name, n, first_stride in (("encoder2", 3, 1), ("encoder3", 4, 2), ("encoder4", 6, 2),
                                  ("encoder5", 3, 2)):
        for b in range(n):
            y = _bottleneck_tv(sd, f"{name}.{b}", y, first_stride if b == 0 else 1, tr)
        feats.append(y)
    e2, e3, e4, e5 = feats
    d = _decoder_block(sd, "decoder5", e5, e4, tr)
    d = _decoder_block(sd, "decoder4", d, e3, tr)
    d = _decoder_block(sd, "decoder3", d, e2, tr)
    d = _decoder_block(sd, "decoder2", d, e1, tr)
    d = F.conv_transpose2d(d, sd["decoder1.0.weight"], sd["decoder1.0.bias"], stride=2)
    d = _relu(_bn(sd, "decoder1.2", _conv(sd, "decoder1.1", d, 1, 1), tr))
    return _conv(sd, "out", d)


def _basic_block_tv(sd, p, x, stride, tr):
    """torchvision BasicBlock: conv3x3(stride)-BN-ReLU-conv3x3-BN + shortcut (downsample.{0,1})."""
    idn = x
    if (p + ".downsample.0.weight") in sd:
        idn = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride, 0), tr)
    y = _relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 1), tr))
    y = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", y, 1, 1), tr)
    return _relu(y + idn)


def _resnet_tv(sd, x, training, block, counts, drop_mask):
    """torchvision ResNet (the hub models helpers.py:158-161 asks for): one bn1, global AVERAGE pool — parity
    unpinned (restated from torchvision's public architecture; torchvision is absent here)."""
    tr = training
    x = _relu(_bn(sd, "bn1", _conv(sd, "conv1", x, 2, 3), tr))
    x = _maxpool(x, 3, 2, 1)
    for li, n in enumerate(counts, start=1):
        for b in range(n):
            x = block(sd, f"layer{li}.{b}", x, 2 if (li > 1 and b == 0) else 1, tr)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return _head(sd, "fc", x, tr, drop_mask)


def resnet18_tv(sd, x, training=False, drop_mask=None):
    return _resnet_tv(sd, x, training, _basic_block_tv, (2, 2, 2, 2), drop_mask)


def resnet50_tv(sd, x, training=False, drop_mask=None):
    return _resnet_tv(sd, x, training, _bottleneck_tv, (3, 4, 6, 3), drop_mask)


NETS = {
    "resnet18_tv": resnet18_tv,
    "resnet50_tv": resnet50_tv,
    "AttentionUNet": attention_unet,
    "R2U_Net": r2u_net,
    "R2AttU_Net": r2attu_net,
    "ResNet18": resnet18,
    "ResNet50": resnet50,
    "VGG16": vgg16,
    "VGG19": vgg19,
    "ResNetUnet": resnet_unet,
    "VGG16_BN": vgg16_bn,
    "VGG19_BN": vgg19_bn,
}


# ----------------------------------------------------------------------------
# state-dict construction (shapes only; values come from an initialiser)
# ----------------------------------------------------------------------------
class _Spec:
    def __init__(self):
        self.entries = OrderedDict()     # key -> (shape, kind, fan_in)

    def conv(self, p, ci, co, k, bias=True):
        self.entries[p + ".weight"] = ((co, ci, k, k), "w", ci * k * k)
        if bias:
            self.entries[p + ".bias"] = ((co,), "b", ci * k * k)

    def convT(self, p, ci, co, k):
        # torch fan_in for ConvTranspose2d weight [ci, co, k, k] is co*k*k
        self.entries[p + ".weight"] = ((ci, co, k, k), "w", co * k * k)
        self.entries[p + ".bias"] = ((co,), "b", co * k * k)

    def bn(self, p, c):
        self.entries[p + ".weight"] = ((c,), "gamma", 0)
        self.entries[p + ".bias"] = ((c,), "beta", 0)
        self.entries[p + ".running_mean"] = ((c,), "rm", 0)
        self.entries[p + ".running_var"] = ((c,), "rv", 0)
        self.entries[p + ".num_batches_tracked"] = ((), "nbt", 0)

    def linear(self, p, i, o):
        self.entries[p + ".weight"] = ((o, i), "w", i)
        self.entries[p + ".bias"] = ((o,), "b", i)


def _spec_double_conv(s, p, ci, co):
    s.conv(p + ".0", ci, co, 3); s.bn(p + ".1", co)
    s.conv(p + ".3", co, co, 3); s.bn(p + ".4", co)


def _spec_up(s, p, ci, co):
    s.conv(p + ".up.1", ci, co, 3); s.bn(p + ".up.2", co)


def _spec_gate(s, p, f, fi):
    s.conv(p + ".W_g.0", f, fi, 1); s.bn(p + ".W_g.1", fi)
    s.conv(p + ".W_x.0", f, fi, 1); s.bn(p + ".W_x.1", fi)
    s.conv(p + ".psi.0", fi, 1, 1); s.bn(p + ".psi.1", 1)


def _spec_rrcnn(s, p, ci, co):
    # registration order in the reference: RCNN first, then conv_1x1 (R2AttU_Net.py:50-54)
    for r in (0, 1):
        s.conv(f"{p}.RCNN.{r}.conv.0", co, co, 3); s.bn(f"{p}.RCNN.{r}.conv.1", co)
    s.conv(p + ".conv_1x1", ci, co, 1)


def spec(name, num_classes=3, head_dropout=False, in_channels=3, out_channels=1):
    """``in_channels`` / ``out_channels``: constructor arguments of the R2 nets (R2U_Net.py:51, R2AttU_Net.py:89); AttentionUNet
    honours only ``out_channel`` — its first block is hard-coded to 3 input channels (AttentionUNet.py:62)."""
    s = _Spec()
    w = (64, 128, 256, 512, 1024)
    if name == "AttentionUNet":
        ci = 3
        for i, c in enumerate(w, start=1):
            _spec_double_conv(s, f"conv{i}", ci, c); ci = c
        for lvl in (5, 4, 3, 2):
            c = w[lvl - 2]
            _spec_up(s, f"up{lvl}", 2 * c, c)
            _spec_gate(s, f"att{lvl}", c, c // 2)
            _spec_double_conv(s, f"up_conv{lvl}", 2 * c, c)
        s.conv("out", 64, out_channels, 1)
    elif name in ("R2U_Net", "R2AttU_Net"):
        ci = in_channels
        for i, c in enumerate(w, start=1):
            _spec_rrcnn(s, f"RRCNN{i}", ci, c); ci = c
        for lvl in (5, 4, 3, 2):
            c = w[lvl - 2]
            _spec_up(s, f"up{lvl}", 2 * c, c)
            if name == "R2AttU_Net":
                _spec_gate(s, f"att{lvl}", c, c // 2)
            _spec_rrcnn(s, f"up_RRCNN{lvl}", 2 * c, c)
        s.conv("conv_1x1", 64, out_channels, 1)
    elif name in ("ResNet18", "ResNet50"):
        s.conv("conv1", 3, 64, 7, bias=False); s.bn("bn1", 64)
        cin = 64
        if name == "ResNet18":
            for li, (c, n) in enumerate(((64, 2), (128, 2), (256, 2), (512, 2)), start=1):
                for b in range(n):
                    p = f"layer{li}.{b}"; stride = 2 if (li > 1 and b == 0) else 1
                    s.conv(p + ".conv1", cin, c, 3, False); s.conv(p + ".conv2", c, c, 3, False)
                    s.bn(p + ".bn1", c); s.bn(p + ".bn2", c)
                    if stride != 1 or cin != c:
                        s.conv(p + ".identity.0", cin, c, 1, False); s.bn(p + ".identity.1", c)
                    cin = c
            feat = 512
        else:
            for li, (c, n) in enumerate(((256, 3), (512, 4), (1024, 6), (2048, 3)), start=1):
                for b in range(n):
                    p = f"layer{li}.{b}"; stride = 2 if (li > 1 and b == 0) else 1
                    s.conv(p + ".conv1", cin, c // 4, 1, False)
                    s.conv(p + ".conv2", c // 4, c // 4, 3, False)
                    s.conv(p + ".conv3", c // 4, c, 1, False)
                    s.bn(p + ".bn1", c // 4); s.bn(p + ".bn2", c // 4); s.bn(p + ".bn3", c)
                    if stride != 1 or cin != c:
                        s.conv(p + ".identity.0", cin, c, 1, False); s.bn(p + ".identity.1", c)
                    cin = c
            feat = 2048
        s.linear("fc.1" if head_dropout else "fc", feat, num_classes)
    elif name in ("VGG16", "VGG19"):
        cfg = VGG16_CFG if name == "VGG16" else VGG19_CFG
        idx, ci = 0, 3
        for c in cfg:
            if c == "M":
                idx += 1
            else:
                s.conv(f"features.{idx}", ci, c, 3); ci = c; idx += 2
        s.linear("classifier.2", 512, 256)
        s.linear("classifier.5", 256, 256)
        s.linear("classifier.9" if head_dropout else "classifier.8", 256, num_classes)
    elif name in ("VGG16_BN", "VGG19_BN"):
        cfg = VGG16_CFG if name == "VGG16_BN" else VGG19_CFG
        idx, ci = 0, 3
        for c in cfg:
            if c == "M":
                idx += 1
            else:
                s.conv(f"features.{idx}", ci, c, 3); s.bn(f"features.{idx + 1}", c); ci = c; idx += 3
        s.linear("classifier.0", 25088, 4096)
        s.linear("classifier.3", 4096, 4096)
        s.linear("classifier.7" if head_dropout else "classifier.6", 4096, num_classes)
    elif name in ("resnet18_tv", "resnet50_tv"):
        s.conv("conv1", 3, 64, 7, bias=False); s.bn("bn1", 64)
        cin = 64
        exp, counts = (1, (2, 2, 2, 2)) if name == "resnet18_tv" else (4, (3, 4, 6, 3))
        for li, (width, n) in enumerate(zip((64, 128, 256, 512), counts), start=1):
            for b in range(n):
                p = f"layer{li}.{b}"; stride = 2 if (li > 1 and b == 0) else 1
                if exp == 1:
                    s.conv(p + ".conv1", cin, width, 3, False); s.bn(p + ".bn1", width)
                    s.conv(p + ".conv2", width, width, 3, False); s.bn(p + ".bn2", width)
                else:
                    s.conv(p + ".conv1", cin, width, 1, False); s.bn(p + ".bn1", width)
                    s.conv(p + ".conv2", width, width, 3, False); s.bn(p + ".bn2", width)
                    s.conv(p + ".conv3", width, 4 * width, 1, False); s.bn(p + ".bn3", 4 * width)
                if stride != 1 or cin != width * exp:
                    s.conv(p + ".downsample.0", cin, width * exp, 1, False); s.bn(p + ".downsample.1", width * exp)
                cin = width * exp
        s.linear("fc.1" if head_dropout else "fc", 512 * exp, num_classes)
    elif name == "ResNetUnet":
        s.conv("encoder1.0", 3, 64, 7, bias=False); s.bn("encoder1.1", 64)
        cin = 64
        for ename, width, n in (("encoder2", 64, 3), ("encoder3", 128, 4), ("encoder4", 256, 6),
                                ("encoder5", 512, 3)):
            for b in range(n):
                p = f"{ename}.{b}"
                s.conv(p + ".conv1", cin, width, 1, False); s.bn(p + ".bn1", width)
                s.conv(p + ".conv2", width, width, 3, False); s.bn(p + ".bn2", width)
                s.conv(p + ".conv3", width, 4 * width, 1, False); s.bn(p + ".bn3", 4 * width)
                if b == 0:
                    s.conv(p + ".downsample.0", cin, 4 * width, 1, False)
                    s.bn(p + ".downsample.1", 4 * width)
                cin = 4 * width
        for p, tot, co in (("decoder5", 3072, 1024), ("decoder4", 1536, 512),
                           ("decoder3", 768, 256), ("decoder2", 320, 64)):
            _spec_double_conv(s, p + ".basic_block", tot, co)
            s.convT(p + ".up_sample", tot - co, tot - co, 2)
        s.convT("decoder1.0", 64, 32, 2)
        s.conv("decoder1.1", 32, 32, 3); s.bn("decoder1.2", 32)
        s.conv("out", 32, 1, 1)
    else:
        raise KeyError(name)
    return s.entries


def _hash01(n, phase):
    """RNG-free pseudo-random U[0,1): frac(sin(12.9898 i + 78.233 phase) * 43758.5453) in f64."""
    i = torch.arange(n, dtype=torch.float64)
    v = torch.sin(i * 12.9898 + phase * 78.233) * 43758.5453
    return v - torch.floor(v)


def closed_form_state(name, scale=1.0, **kw):
    """Deterministic, RNG-free weights (independent of any library's random stream):
    u = _hash01(i, key_index); conv/linear weights U(+-sqrt(6/fan_in)) (He-uniform, keeps
    activation variance through ReLU stacks), biases U(+-0.05), gamma 1 +- 0.2, beta +-0.1,
    running_mean +-0.05, running_var 1 +- 0.3."""
    sd = OrderedDict()
    for k_i, (key, (shape, kind, fan_in)) in enumerate(spec(name, **kw).items()):
        if kind == "nbt":
            sd[key] = torch.zeros((), dtype=torch.int64)
            continue
        n = int(math.prod(shape)) if shape else 1
        base = 2.0 * _hash01(n, 1.0 + k_i) - 1.0
        if kind == "w":
            v = base * math.sqrt(6.0 / fan_in) * scale
        elif kind == "b":
            v = base * 0.05
        elif kind == "gamma":
            v = 1.0 + 0.2 * base
        elif kind == "beta":
            v = 0.1 * base
        elif kind == "rm":
            v = 0.05 * base
        else:  # rv
            v = 1.0 + 0.3 * base
        sd[key] = v.to(torch.float32).reshape(shape).contiguous()
    return sd


def default_init_state(name, seed=0, **kw):
    """torch-default initialisation (Kaiming-uniform a=sqrt(5) => U(+-1/sqrt(fan_in))
    for weights and biases; BN gamma 1, beta 0, running 0/1) drawn from a seeded
    CPU generator in key order — the build's own init, used for benchmarks."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for key, (shape, kind, fan_in) in spec(name, **kw).items():
        if kind in ("w", "b"):
            bound = 1.0 / math.sqrt(fan_in)
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind in ("gamma", "rv"):
            sd[key] = torch.ones(shape)
        elif kind in ("beta", "rm"):
            sd[key] = torch.zeros(shape)
        else:
            sd[key] = torch.zeros((), dtype=torch.int64)
    return sd


def is_buffer(key):
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def param_keys(sd):
    return [k for k in sd if not is_buffer(k)]


def closed_form_fill(sd, salt=0.0):
    """RNG-free values for an arbitrary module ``state_dict`` (block-level fixtures): returns a new
    dict with the same keys/shapes.  >1-D tensors ~ U(+-1.5*sqrt(3/fan_in)); BN gamma 1+-0.2;
    biases/beta +-0.1; running_mean +-0.05; running_var 1+-0.3; counters 0."""
    out = OrderedDict()
    for i, (k, v) in enumerate(sd.items()):
        if not v.is_floating_point():
            out[k] = torch.zeros_like(v)
            continue
        u = (2.0 * _hash01(v.numel(), salt + 3.0 + i) - 1.0).reshape(v.shape)
        if v.dim() > 1:
            fan_in = v[0].numel()
            val = u * 1.5 * math.sqrt(3.0 / fan_in)
        elif k.endswith("running_var"):
            val = 1.0 + 0.3 * u
        elif k.endswith("running_mean"):
            val = 0.05 * u
        elif k.endswith("weight"):
            val = 1.0 + 0.2 * u
        else:
            val = 0.1 * u
        out[k] = val.to(v.dtype)
    return out


def closed_form_tensor(shape, salt):
    """RNG-free ~N(0,1)-ish tensor (sum of three hashed uniforms, centred and scaled)."""
    n = int(math.prod(shape))
    u = _hash01(n, salt) + _hash01(n, salt + 0.37) + _hash01(n, salt + 0.71)
    return ((u - 1.5) * 2.0).float().reshape(shape)
