"""CPU: host-side logic of the launch-plan builder (no kernels run): state_dict compatibility with
the reference layouts, ABI arity of every emitted launch, gradient bookkeeping (first write /
accumulate), frozen-parameter pruning, eval plans."""
import pytest
import torch

from mi355 import graph
from mi355.lib import lib, available
from oracle import nets

pytestmark = pytest.mark.skipif(not available(), reason="libmi355conv.so not built")


def _models():
    from models.segmentation_models.AttentionUNet import AttentionUNet
    from models.segmentation_models.R2AttU_Net import R2AttU_Net
    from models.segmentation_models.R2U_Net import R2U_Net
    from models.classification_models.ResNet import ResNet18, ResNet50
    from models.classification_models.VGG import VGG16, VGG19, VGG16_BN, VGG19_BN
    from models.classification_models.TorchvisionResNet import resnet18, resnet50
    return {"resnet18_tv": (lambda: resnet18(3), (2, 3, 64, 64)), "resnet50_tv": (lambda: resnet50(3), (2, 3, 64, 64)),
            "AttentionUNet": (AttentionUNet, (2, 3, 32, 32)), "R2AttU_Net": (R2AttU_Net, (1, 3, 32, 32)),
            "R2U_Net": (R2U_Net, (1, 3, 32, 32)), "ResNet18": (lambda: ResNet18(3), (2, 3, 64, 64)),
            "ResNet50": (lambda: ResNet50(3), (2, 3, 64, 64)), "VGG16": (lambda: VGG16(3), (2, 3, 32, 32)),
            "VGG19": (lambda: VGG19(3), (2, 3, 32, 32)), "VGG16_BN": (lambda: VGG16_BN(3), (2, 3, 32, 32)),
            "VGG19_BN": (lambda: VGG19_BN(3), (2, 3, 32, 32))}


@pytest.mark.parametrize("name", list(_models()))
def test_state_dict_layout_matches_reference(name):
    ctor, _ = _models()[name]
    sd = ctor().state_dict()
    spec = nets.spec(name)
    assert set(sd) == set(spec)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(spec[k][0]), k


@pytest.mark.parametrize("name", ["AttentionUNet", "R2AttU_Net", "ResNet18", "VGG16"])
def test_plan_structure(name):
    ctor, shape = _models()[name]
    net = ctor().train()
    eng = net.engine
    eng.flatten()
    plan = eng.plan_for(shape, True, True, torch.float32)
    fwd, bwd = plan.bind(0)                     # resolves every pointer and checks ABI arity; launches nothing
    assert len(fwd) == plan.n_launches[0] and len(bwd) == plan.n_launches[1] and len(bwd) > 0
    params = list(net.parameters())
    assert {id(p) for p in plan.grad_params} == {id(p) for p in params}
    # each parameter: exactly one overwriting (beta/acc == 0) gradient write, the rest accumulate
    first, later = {}, {}
    for l in plan.bwd:
        for i, a in enumerate(l.args):
            if isinstance(a, graph.GRef):
                beta = l.args[-1] if l.name in ("mi355_conv2d_wgrad_reduce", "mi355_colsum_finalize", "mi355_bn_bwd_finalize", "mi355_bn_bwd_finalize_at") else \
                    (l.args[-2] if l.name == "mi355_linear_bwd" else None)          # (..., beta, scratch)
                assert beta is not None, l.name
                d = first if beta == 0.0 else later
                d[id(a.param)] = d.get(id(a.param), 0) + 1
    zero = {id(p) for p in plan.zero_grad_params}        # conv biases in front of a train-mode BN: gradient is exactly 0
    names_of = {id(p): k for k, p in net.named_parameters()}
    assert all(names_of[i].endswith(".bias") for i in zero)
    for p in params:
        assert first.get(id(p), 0) >= 1 or id(p) in zero, "parameter never written"
    if name == "R2AttU_Net":
        assert sum(later.values()) > 0            # shared recurrent weights accumulate over 6 applications
    elif name == "ResNet18":
        assert sum(later.values()) == 2           # bn1 (gamma, beta) is applied twice (ResNet.py:130,134)
    else:
        assert sum(later.values()) == 0
    # eval plan: no backward, running-stat coefficients instead of batch statistics
    net.eval()
    ev = eng.plan_for(shape, False, False, torch.float32)
    assert ev.n_launches[1] == 0
    names = {l.name for l in ev.fwd}
    assert "mi355_bn_stats" not in names and ("mi355_bn_eval_coeffs" in names or name == "VGG16")


def test_frozen_backbone_prunes_backward():
    """Stage 1 of the classification protocol (helpers.py:258-283): only the head trains."""
    from models.classification_models.ResNet import ResNet18
    from utils.helpers import add_dropout_to_fc
    net = ResNet18(num_classes=1000)
    head = add_dropout_to_fc(net, p=0.5)
    assert head == "fc"
    for p in net.parameters():
        p.requires_grad = False
    for p in net.fc.parameters():
        p.requires_grad = True
    net.train()
    net.engine.flatten()
    plan = net.engine.plan_for((2, 3, 64, 64), True, True, torch.float32)
    assert {id(p) for p in plan.grad_params} == {id(p) for p in net.fc.parameters()}
    assert all(l.name in ("mi355_linear_bwd", "mi355_dropout_bwd") for l in plan.bwd), {l.name for l in plan.bwd}
    assert any(l.name == "mi355_dropout_fwd" for l in plan.fwd)


def test_concat_is_free_and_slices_share_storage():
    from models.segmentation_models.AttentionUNet import AttentionUNet
    net = AttentionUNet().train()
    net.engine.flatten()
    plan = net.engine.plan_for((1, 3, 32, 32), True, True, torch.bfloat16)
    assert not any("cat" in l.name for l in plan.fwd)
    # UpConv's bn_act and the gate multiply write into the two halves of one buffer (ld = 2C)
    gate = [l for l in plan.fwd if l.name == "mi355_gate_mul_fwd"]
    assert len(gate) == 4 and all(l.args[6] == 2 * l.args[8] for l in gate)   # ldy == 2*C


def test_maxpool_rides_in_the_batchnorm_apply_pass():
    """AttentionUNet.py:61,89-95: the four MaxPool2d(2, 2) read the activation a BatchNorm apply pass has just written — the
    plan emits ONE mi355_bn_act_pool2 per encoder level instead of mi355_bn_act + mi355_maxpool_fwd; the backward still sees
    the pooling (mi355_maxpool_bwd) and Plan.acts still lists it (the kink replay of tests/test_gpu_kinks.py)."""
    from models.segmentation_models.AttentionUNet import AttentionUNet
    net = AttentionUNet().train()
    net.engine.flatten()
    plan = net.engine.plan_for((2, 3, 64, 64), True, True, torch.bfloat16)
    names = [l.name for l in plan.fwd]
    pooled = [l for l in plan.fwd if l.name == "mi355_bn_act_pool2" and l.args[6] is not None]
    assert len(pooled) == 4 and "mi355_maxpool_fwd" not in names
    # (the other plain apply passes on even images run the same window-ordered kernel without a pooled output)
    assert all(l.args[6] is None for l in plan.fwd if l.name == "mi355_bn_act_pool2" and l not in pooled) and "mi355_bn_act" not in names
    assert sum(a[0] == "pool" for a in plan.acts) == 4
    # ... and the pooling's gradient rides in the two BatchNorm backward passes of the layer that produced the pooled activation
    # (no mi355_maxpool_bwd pass over its gradient)
    bnames = [l.name for l in plan.bwd]
    assert bnames.count("mi355_maxpool_bwd") == 0 and bnames.count("mi355_bn_bwd_reduce_pool2") == 4 \
        and bnames.count("mi355_bn_bwd_apply_pool2") == 4
    # a width the window-ordered pass does not cover (W = 40: not a power-of-two multiple of 64 pixels) keeps the separate pass
    plan = net.engine.plan_for((1, 3, 80, 80), True, True, torch.bfloat16)
    bnames = [l.name for l in plan.bwd]
    assert bnames.count("mi355_maxpool_bwd") + bnames.count("mi355_bn_bwd_reduce_pool2") == 4 and "mi355_maxpool_bwd" in bnames
    for i, nm in enumerate(bnames):
        if nm == "mi355_bn_bwd_reduce_pool2":
            assert bnames[i + 1] == "mi355_bn_bwd_finalize" and bnames[i + 2] == "mi355_bn_bwd_apply_pool2"
    plan.bind(0)                                   # ABI arity of the new entry point


def test_no_cpu_fallback():
    from models.segmentation_models.AttentionUNet import AttentionUNet
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        AttentionUNet()(torch.zeros(1, 3, 32, 32))


def test_tester_reports_match_the_reference_text(tmp_path, capsys):
    """utils.tester.print_summary / save_results_to_csv against the text and the CSV files the REFERENCE's functions
    (tester.py:738-876, run by oracle/make_golden.py) produced for the same result dictionaries: both tables, the best-model
    lines, the family split by model name, the columns dropped from the classification CSV, pandas' float formatting, and the
    empty / one-family branches."""
    import os
    import numpy as np
    from utils import tester
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "tester.npz"), allow_pickle=False)
    cls = {k: float(z[f"clsloop/{k}"]) for k in ("accuracy", "precision", "recall", "f1")}
    cls.update({k: z[f"clsloop/{k}"] for k in ("precision_per_class", "recall_per_class", "f1_per_class", "confusion_matrix")})
    seg = dict(zip([str(k) for k in z["seg_keys"]], [float(v) for v in z["segloop/metrics"]]))
    res = {"ResNet18": cls, "VGG16": {"accuracy": 93.0, "precision": 92.0, "recall": 93.0, "f1": 92.5}, "AttentionUNet": seg,
           "R2Unet": {"iou": 79.0, "dice": 87.25, "pixel_accuracy": 96.5, "precision": 88.0, "recall": 86.5, "f1": 87.24}}

    def said(fn, *a):
        fn(*a)
        return capsys.readouterr().out.replace(str(tmp_path), "<dir>")
    assert said(tester.print_summary, res) == str(z["report/summary"])
    assert said(tester.print_summary, {}) == str(z["report/summary_empty"])
    assert said(tester.print_summary, {"AttentionUNet": seg}) == str(z["report/summary_seg_only"])
    c, s_ = tmp_path / "c.csv", tmp_path / "s.csv"
    assert said(tester.save_results_to_csv, res, str(c), str(s_)) == str(z["report/csv_stdout"])
    assert c.read_text() == str(z["report/csv_cls"]) and s_.read_text() == str(z["report/csv_seg"])
    assert said(tester.save_results_to_csv, {"AttentionUNet": seg}, str(c), str(s_)) == str(z["report/csv_stdout_seg_only"])
    assert said(tester.save_results_to_csv, {}, str(c), str(s_)) == str(z["report/csv_stdout_empty"])


def test_recurrent_convs_get_one_weight_gradient_launch(monkeypatch):
    """A convolution applied six times (Recurrent_block, t = 5) on a level the nine-tap kernel serves emits ONE
    mi355_conv2d_wgrad_multi over its six (x, dy) pairs and ONE overwriting reduce; smaller levels and
    MI355_WGRAD_MULTI=0 keep one launch per application (first overwrites, five accumulate)."""
    from models.segmentation_models.R2U_Net import R2U_Net

    def plan_of(multi):
        monkeypatch.setenv("MI355_WGRAD_MULTI", multi)
        net = R2U_Net().train()
        net.engine.flatten()
        plan = net.engine.plan_for((2, 3, 32, 32), True, True, torch.bfloat16)
        plan.bind(0)                                 # ABI arity of every launch, the new entry point included
        return net, plan

    net, plan = plan_of("1")
    multis = [l for l in plan.bwd if l.name == "mi355_conv2d_wgrad_multi"]
    assert multis and all(l.args[12] == 6 for l in multis)              # napp
    assert all(sum(a is not None for a in l.args[:12]) == 12 for l in multis)
    for l in multis:                                                     # six distinct inputs, six distinct gradients
        assert len({id(a) for a in l.args[0:12:2]}) == 6 and len({id(a) for a in l.args[1:12:2]}) == 6
    # levels: 32x32 (rows of 32) and 16x16 (two images at a time) are served; 8x8 and below are not
    served = {(l.args[20], l.args[21]) for l in multis}                  # (Ho, Wo)
    assert served == {(32, 32), (16, 16)}
    # every parameter still gets exactly one overwriting gradient write
    betas = {}
    for l in plan.bwd:
        if l.name == "mi355_conv2d_wgrad_reduce":
            betas.setdefault(id(l.args[2].param), []).append(l.args[-1])
    assert len(betas) >= 30 and all(b.count(0.0) == 1 for b in betas.values())      # (the Co = 1 head goes through rowdot_bwd)
    n_reduce_multi = sum(l.name == "mi355_conv2d_wgrad_reduce" for l in plan.bwd)
    _, plan0 = plan_of("0")
    assert not any(l.name == "mi355_conv2d_wgrad_multi" for l in plan0.bwd)
    n_reduce_single = sum(l.name == "mi355_conv2d_wgrad_reduce" for l in plan0.bwd)
    assert n_reduce_single - n_reduce_multi == 5 * len(multis)


def test_small_grid_tile_rule():
    """csrc/conv_igemm.hip small_grid_tile_n / dma_tile_n (host code; 256 CUs assumed without a device): the widest output-channel
    tile that divides Co, narrower when that grid would be at most half a round of workgroups."""
    assert lib.mi355_conv2d_igemm_dma_tile(32, 8, 8, 512, 512) == 32            # ResNet-50 layer4 at batch 32: 16 x 4 wide tiles
    assert lib.mi355_conv2d_igemm_dma_tile(32, 16, 16, 1024, 256) == 64
    assert lib.mi355_conv2d_igemm_dma_tile(32, 32, 32, 512, 256) == 128         # 256 row tiles: the wide tile stays
    assert lib.mi355_conv2d_igemm_dma_tile(2, 8, 8, 96, 64) == 64               # Ci % 64 != 0: no 32-wide instance
    assert lib.mi355_conv2d_igemm_generic_tile(8, 8, 8, 512) == 32              # ResNet-18 layer4 at batch 8
    assert lib.mi355_conv2d_igemm_generic_tile(32, 256, 256, 128) == 128
    assert lib.mi355_conv2d_igemm_generic_tile(2, 8, 8, 96) == 32               # 96 channels: 32-wide tiles anyway


def test_recurrent_block_input_gradient_is_summed_once(monkeypatch):
    """R2AttU_Net (t = 5: six applications per recurrent block, five of which add the block input, R2AttU_Net.py:41-44): with
    DEFER_POST every block's backward has ONE mi355_bn_bwd_apply_post4 carrying four earlier incoming gradients of the pitch of the
    current one, and no plain apply pass of a recurrent application carries a post-activation operand; without it there is no post4
    launch and five passes per block read-modify-write the operand's gradient.  ABI arity checked by bind()."""
    from models.segmentation_models.R2AttU_Net import R2AttU_Net

    def plan_of(defer):
        monkeypatch.setattr(graph, "DEFER_POST", defer)
        net = R2AttU_Net().train()
        net.engine.flatten()
        plan = net.engine.plan_for((1, 3, 32, 32), True, True, torch.bfloat16)
        plan.bind(0)
        return plan

    on, off = plan_of(True), plan_of(False)
    post4 = [l for l in on.bwd if l.name == "mi355_bn_bwd_apply_post4"]
    assert len(post4) == 18                                   # 9 RRCNN blocks x 2 recurrent blocks
    for l in post4:
        ex = l.args[17:21]
        assert all(e is not None for e in ex) and len({id(e) for e in ex} | {id(l.args[0])}) == 5      # four earlier gradients + this one
        assert all(e.ld == l.args[21] for e in ex) and l.args[14] is not None                          # one pitch; the operand's gradient
    plain_on = [l for l in on.bwd if l.name == "mi355_bn_bwd_apply"]
    plain_off = [l for l in off.bwd if l.name == "mi355_bn_bwd_apply"]
    assert not any(l.name == "mi355_bn_bwd_apply_post4" for l in off.bwd)
    with_post = lambda ls: sum(l.args[16] is not None for l in ls)                                       # dpost argument
    assert with_post(plain_off) - with_post(plain_on) == 18 * 5 and len(plain_off) == len(plain_on) + 18


@pytest.mark.parametrize("name", ["ResNetUnet", "AttentionUNet", "R2AttU_Net", "ResNet50", "VGG16_BN"])
def test_slab_workspace_is_owned_by_one_stream(name):
    """The split-K slab workspace Ws('bytes') is shared by every weight-gradient launch of a plan; the side stream
    only waits for the main stream at a fork, so a main-stream user of the same workspace would race with it
    (ConvTranspose2d of ResnetUnet.py:21,51 once did).  All users must sit on the same stream."""
    if name == "ResNetUnet":
        from models.segmentation_models.ResnetUnet import ResNetUnet
        ctor, shape = (lambda: ResNetUnet(freeze=False)), (1, 3, 64, 64)
    else:
        ctor, shape = _models()[name]
    net = ctor().train()
    net.engine.flatten()
    plan = net.engine.plan_for(shape, True, True, torch.bfloat16)
    def uses(l, kind):       # a workspace argument is the Ws itself or a late-bound (Ws, byte offset) pair
        return any((isinstance(a, graph.Ws) and a.kind == kind) or
                   (isinstance(a, tuple) and len(a) == 2 and isinstance(a[0], graph.Ws) and a[0].kind == kind) for a in l.args)
    users = [l for l in plan.bwd if uses(l, "bytes")]
    assert users and {l.side for l in users} == {True}, {(l.name, l.side) for l in users}
    # and the other shared workspace never appears on the side stream (the psi / head weight-gradient folds that run there read
    # buffers of their own)
    assert not any(l.side for l in plan.bwd if uses(l, "f32")), [l.name for l in plan.bwd if l.side and uses(l, "f32")]
    if name == "AttentionUNet":
        assert sum(l.side and l.name == "mi355_colsum_finalize" for l in plan.bwd) == 10      # psi weight + bias of four gates, the logit head's


def test_torchvision_resnet_layouts_have_the_published_sizes():
    """helpers.py:158-161 trains torchvision's resnet18 / resnet50 when the hub is reachable: 11 689 512 / 25 557 032
    parameters at 1000 classes, global AVERAGE pool, one bn1 application; the factory swaps the head like the reference."""
    from models.classification_models import TorchvisionResNet as tv
    from utils.helpers import get_class_model
    assert sum(p.numel() for p in tv.resnet18().parameters()) == 11_689_512
    assert sum(p.numel() for p in tv.resnet50().parameters()) == 25_557_032
    m, head = get_class_model("resnet50", hub=True)
    assert head == "fc" and isinstance(m.fc[0], torch.nn.Dropout) and m.fc[1].out_features == 3
    assert "layer1.0.downsample.0.weight" in m.state_dict() and "layer1.1.downsample.0.weight" not in m.state_dict()
    m.train()
    m.engine.flatten()
    plan = m.engine.plan_for((2, 3, 64, 64), True, True, torch.float32)
    pools = [l for l in plan.fwd if l.name == "mi355_global_pool_fwd"]
    assert len(pools) == 1 and pools[0].args[-2] == 0                      # is_max == 0: average
    assert sum(l.name == "mi355_bn_finalize" for l in plan.fwd) == 53      # 49 block BNs + 4 downsample, bn1 once
    local, _ = get_class_model("resnet50")
    assert "layer1.0.identity.0.weight" in local.state_dict()              # default: the reference's offline fallback classes


@pytest.mark.parametrize("freeze", [True, False])
def test_resnet_unet_layout_and_freeze_pattern_match_the_reference_class(freeze):
    """models/segmentation_models/ResnetUnet.py against the REFERENCE's ResNetUnet (ResnetUnet.py:29-83 instantiated over a
    plain-torch container of torchvision's ResNet-50 layout, tests/golden/model_ResNetUnet.npz): the same state_dict keys in the
    same order, the same parameter order, the same requires_grad pattern after `_freeze_backbone` (ResnetUnet.py:60-66) — and the
    launch plan gives a gradient to exactly the parameters the reference's backward gave one (frozen encoder: its weight- and
    data-gradient launches are pruned, its BatchNorm statistic launches stay in the forward)."""
    import os
    import numpy as np
    from models.segmentation_models.ResnetUnet import ResNetUnet
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_ResNetUnet.npz"), allow_pickle=False)
    tag = "frozen" if freeze else "unfrozen"
    net = ResNetUnet(n_classes=1, freeze=freeze)
    assert list(net.state_dict().keys()) == [str(k) for k in z["state_keys"]]
    names = [k for k, _ in net.named_parameters()]
    assert names == [str(k) for k in z[f"{tag}/param_names"]]
    assert np.array_equal(np.array([p.requires_grad for _, p in net.named_parameters()]), z[f"{tag}/requires_grad"])
    net.train()
    net.engine.flatten()
    plan = net.engine.plan_for((2, 3, 64, 64), True, True, torch.float32)
    got = {id(p) for p in plan.grad_params}
    want = {id(p) for (k, p), h in zip(net.named_parameters(), z[f"{tag}/has_grad"]) if h}
    assert got == want
    n_bn = sum(isinstance(mod, torch.nn.BatchNorm2d) for mod in net.modules())
    assert sum(l.name == "mi355_bn_finalize" for l in plan.fwd) == n_bn          # every BatchNorm runs in train mode, frozen or not
