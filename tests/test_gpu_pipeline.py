"""-m gpu: the joint classify -> segment-if-COVID path (utils/pipeline.py:324-418) batched on the GPU against
the per-image oracle (oracle/pipeline.py) on the same weights and images.  fp32 compute: class decisions
identical, confidences within 1e-3 relative, masks identical except for <= 0.1 % of pixels (logit at the threshold);
bf16 / fp16: decisions identical wherever the oracle's top-2 logit margin exceeds 1.0 (raw logits are O(40), so
that is 8 bf16 ulps), mask disagreement <= 0.5 % of pixels."""
import pytest
import torch

from oracle import nets, pipeline as opipe

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


HEAD_BIAS = {"ResNet18": "fc.1.bias", "VGG16_BN": "classifier.7.bias"}


def _he(sd, linear=False):
    """default init has gain 1/sqrt(3); eval-mode BN with fresh running statistics is the identity, so rescale the
    convolutions (and, for the three-layer VGG head, the Linears) to He gain to keep activations O(1) through the depth"""
    for v in sd.values():
        if v.dim() == 4 or (linear and v.dim() == 2):
            v.mul_(6 ** 0.5)
    return sd


def _fixture(cls_name="ResNet18", hw=64):
    cls_sd = _he(nets.default_init_state(cls_name, seed=3, num_classes=3, head_dropout=True), linear=cls_name == "VGG16_BN")
    seg_sd = _he(nets.default_init_state("AttentionUNet", seed=4))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 3, hw, hw, generator=g)
    if cls_name == "VGG16_BN":       # a random VGG sees pure noise images as one image: vary contrast and colour cast per sample
        x = x * (0.3 + 1.4 * torch.rand(16, 1, 1, 1, generator=g)) + 1.5 * torch.randn(16, 3, 1, 1, generator=g)
    fwd = nets.NETS[cls_name]
    with torch.no_grad():        # centre the logits over the batch: the three classes all occur
        cls_sd[HEAD_BIAS[cls_name]] = cls_sd[HEAD_BIAS[cls_name]] - fwd({k: v.clone() for k, v in cls_sd.items()}, x, False).mean(0)
        z = fwd({k: v.clone() for k, v in cls_sd.items()}, x, False)
    top = z.sort(1, descending=True).values
    return cls_sd, seg_sd, x, top[:, 0] - top[:, 1]


def _models(dtype, cls_sd, seg_sd, cls_name="ResNet18"):
    from models.classification_models.ResNet import ResNet18
    from models.classification_models.VGG import VGG16_BN
    from models.segmentation_models.AttentionUNet import AttentionUNet
    from utils.helpers import add_dropout_to_fc
    cm = {"ResNet18": ResNet18, "VGG16_BN": VGG16_BN}[cls_name](num_classes=3)
    add_dropout_to_fc(cm)
    cm.load_state_dict(cls_sd)
    sm = AttentionUNet()
    sm.load_state_dict(seg_sd)
    cm.compute_dtype = sm.compute_dtype = dtype
    return cm, sm


# ResNet18: raw logits are O(40), a top-2 margin of 1.0 is 8 bf16 ulps; VGG16_BN (config C5's classifier, fp16): logits O(5)
@pytest.mark.parametrize("cls_name,dtype,min_margin", [
    ("ResNet18", torch.float32, 0.0), ("ResNet18", torch.bfloat16, 1.0), ("ResNet18", torch.float16, 1.0),
    ("VGG16_BN", torch.float32, 0.0), ("VGG16_BN", torch.float16, 0.5), ("VGG16_BN", torch.bfloat16, 0.8),
])
def test_joint_pipeline_matches_per_image_oracle(cls_name, dtype, min_margin):
    from utils.pipeline import JointPipeline
    cls_sd, seg_sd, x, margin = _fixture(cls_name)
    cm, sm = _models(dtype, cls_sd, seg_sd, cls_name)
    ref = opipe.process_batch(cls_name, cls_sd, "AttentionUNet", seg_sd, x)
    n_pos = sum(r[2] is not None for r in ref)
    assert 2 <= n_pos <= len(ref) - 2, "degenerate fixture: the batch must mix COVID and non-COVID predictions"
    pipe = JointPipeline(cm, sm, device=DEV, bucket=4)
    got = pipe.process_batch(x)
    sure = margin > min_margin
    assert int(sure.sum()) >= 10
    tol = 1e-3 if dtype == torch.float32 else 0.25
    checked_masks = 0
    for i, ((p, c, m), (pr, cr, mr)) in enumerate(zip(got, ref)):
        if not bool(sure[i]):
            continue
        assert p == pr, (i, p, pr, float(margin[i]))
        assert abs(c - cr) <= tol * cr, (i, c, cr)
        assert (m is None) == (mr is None)
        if m is not None:
            diff = float((m != mr).float().mean())
            assert diff <= (1e-3 if dtype == torch.float32 else 5e-3), diff
            checked_masks += 1
    assert checked_masks >= 2
    # masks of non-kept samples are all zero; another batch size adds its own plans
    r = pipe.predict(x[:3])
    assert r["masks"][~r["segmented"]].sum() == 0
    # no segmentation model: classification only (pipeline.py:343-347 returns None)
    only = JointPipeline(cm, None, device=DEV).process_batch(x[:4])
    assert all(o[2] is None for o in only) and [o[0] for o in only] == [g_[0] for g_ in got[:4]]


def test_joint_pipeline_c5_shape_fp16_agrees_with_fp32():
    """Config C5 at its own shape — vgg16_bn + AttentionUNet, 512x512, batch 16, fp16 — against the fp32 HIP path on the same
    weights and images (the per-image CPU oracle would take minutes at this size; both HIP precisions are pinned to it at
    64x64 above): identical decisions where the fp32 margin is clear, masks <= 0.5 % of pixels apart, the segmenter only ran
    on the kept samples."""
    from utils.pipeline import JointPipeline
    cls_sd = _he(nets.default_init_state("VGG16_BN", seed=3, num_classes=3, head_dropout=True), linear=True)
    seg_sd = _he(nets.default_init_state("AttentionUNet", seed=4))
    g = torch.Generator().manual_seed(6)
    x = torch.randn(16, 3, 512, 512, generator=g) * (0.3 + 1.4 * torch.rand(16, 1, 1, 1, generator=g)) + 1.5 * torch.randn(16, 3, 1, 1, generator=g)
    out = {}
    for dtype in (torch.float32, torch.float16):
        cm, sm = _models(dtype, cls_sd, seg_sd, "VGG16_BN")
        cm = cm.to(DEV).eval()
        if dtype == torch.float32:
            with torch.no_grad():
                z = cm(x.to(DEV)).float()
                cm.classifier[7].bias -= z.mean(0)           # centre the logits: the three classes all occur
                cls_sd = {k: v.detach().cpu().clone() for k, v in cm.state_dict().items()}
                z = cm(x.to(DEV)).float().cpu()
            top = z.sort(1, descending=True).values
            margin = top[:, 0] - top[:, 1]
        out[dtype] = JointPipeline(cm, sm, device=DEV, bucket=4).predict(x)
    a, b = out[torch.float32], out[torch.float16]
    sure = (margin > 0.5).to(DEV)
    assert int(sure.sum()) >= 8 and 2 <= int(a["segmented"].sum()) <= 14
    assert torch.equal(a["pred"][sure], b["pred"][sure])
    both = a["segmented"] & b["segmented"] & sure
    assert int(both.sum()) >= 2
    diff = (a["masks"][both] != b["masks"][both]).float().mean()
    assert float(diff) <= 5e-3, float(diff)
    assert b["masks"][~b["segmented"]].sum() == 0


def test_pipeline_glue_kernels():
    from mi355.lib import lib
    g = torch.Generator().manual_seed(1)
    z = torch.randn(37, 3, generator=g)
    z[5] = torch.tensor([1.0, 1.0, 0.0])             # tie: first maximum wins (torch.max)
    pred = torch.empty(37, dtype=torch.int32, device=DEV); conf = torch.empty(37, device=DEV)
    kept = torch.empty(37, dtype=torch.int32, device=DEV); nk = torch.empty(1, dtype=torch.int32, device=DEV)
    lib.mi355_cls_decide(z.to(DEV), 37, 3, 0, pred, conf, kept, nk)
    p = torch.softmax(z, 1)
    cr, ir = p.max(1)
    assert torch.equal(pred.cpu().long(), ir) and torch.allclose(conf.cpu(), cr * 100, rtol=1e-5)
    want = torch.nonzero(ir == 0).flatten()
    assert int(nk) == len(want) and torch.equal(kept[: int(nk)].cpu().long(), want)
    x = torch.randn(6, 3, 8, 8, generator=g).to(DEV)
    idx = torch.tensor([4, 1, 1, 5], dtype=torch.int32, device=DEV)
    y = torch.empty(4, 3, 8, 8, device=DEV)
    lib.mi355_gather_rows(x, idx, 4, 3 * 64, y)
    assert torch.equal(y, x[idx.long()])
    lg = torch.randn(2, 1, 8, 8, generator=g)
    out = torch.zeros(6, 8, 8, dtype=torch.uint8, device=DEV)
    lib.mi355_mask_scatter(lg.to(DEV), idx, 2, 64, 0.5, out)
    exp = torch.zeros(6, 8, 8, dtype=torch.uint8)
    exp[4] = (torch.sigmoid(lg[0, 0]) > 0.5).to(torch.uint8) * 255
    exp[1] = (torch.sigmoid(lg[1, 0]) > 0.5).to(torch.uint8) * 255
    assert torch.equal(out.cpu(), exp)


def test_pipeline_from_png_files(tmp_path):
    """JointPipeline.process_files: PNG files -> native decode -> A.Resize(256) + Normalize on the GPU -> classify -> segment the
    "COVID" ones; equal to process_batch on the oracle's per-sample transform of the PIL-decoded images (pipeline.py:381-398)."""
    Image = pytest.importorskip("PIL.Image")
    import numpy as np
    from oracle import transforms as ot
    from utils.pipeline import JointPipeline
    cls_sd, seg_sd, _, _ = _fixture("ResNet18")
    cm, sm = _models(torch.float32, cls_sd, seg_sd, "ResNet18")
    g = np.random.RandomState(2)
    paths = []
    for i in range(5):
        yy, xx = np.mgrid[0:299, 0:299]
        img = (127 + 80 * np.sin(xx / (9.0 + i)) * np.cos(yy / (6.0 + i)) + g.randint(-10, 10, (299, 299))).clip(0, 255).astype(np.uint8)
        p = str(tmp_path / f"x{i}.png")
        Image.fromarray(img, "L").save(p)
        paths.append(p)
    pipe = JointPipeline(cm, sm, device=DEV, bucket=4)
    got = pipe.process_files(paths, size=64)
    xs = torch.stack([torch.from_numpy(ot.normalize_u8(ot.warp_u8(np.array(Image.open(p).convert("RGB")), ot.resize_matrix(299, 299, 64, 64), 64, 64)))
                      for p in paths])
    want = pipe.process_batch(xs)
    assert [g_[0] for g_ in got] == [w[0] for w in want]
    for (p1, c1, m1), (p2, c2, m2) in zip(got, want):
        assert abs(c1 - c2) <= 1e-2 * c2 and (m1 is None) == (m2 is None)
        if m1 is not None:
            assert float((m1 != m2).float().mean()) <= 5e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_joint_pipeline_matches_reference_pipeline_fixture(dtype):
    """The REFERENCE's `Pipeline._predict_classification` / `_predict_segmentation` (pipeline.py:324-357, run by
    oracle/make_golden.py with its own ResNet18 / AttentionUNet set by hand) against the batched GPU path on the same weights
    and images: class, confidence in percent, and the uint8 mask of every image the decision of `process_image` segments."""
    import os
    import numpy as np
    from oracle import train as otrain
    from utils.pipeline import JointPipeline
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline.npz"), allow_pickle=False)
    cls_sd = nets.closed_form_state("ResNet18", head_dropout=True)
    cls_sd["fc.1.bias"] = cls_sd["fc.1.bias"] - torch.from_numpy(z["bias_shift"])
    cm, sm = _models(dtype, cls_sd, nets.closed_form_state("AttentionUNet"), "ResNet18")
    x, _ = otrain.synthetic_batch(12, 64, seed=21, classes=3)
    got = JointPipeline(cm, sm, device=DEV, bucket=4).process_batch(x)
    masks = np.unpackbits(z["masks"], axis=-1)[..., :int(z["hw"])].astype(np.uint8) * 255
    classes = [str(c) for c in z["classes"]]
    sure = z["logit_margin"] > (0.0 if dtype == torch.float32 else 1.0)
    assert int(sure.sum()) >= 9
    n_masks = 0
    for i, (p, c, m) in enumerate(got):
        if not sure[i]:
            continue
        assert p == classes[int(z["pred"][i])], i
        assert abs(c - float(z["confidence"][i])) <= (1e-3 if dtype == torch.float32 else 0.25) * float(z["confidence"][i])
        assert (m is not None) == (p == "COVID")
        if m is not None:
            bad = int((m.numpy() != masks[i]).sum())
            assert bad <= (int(z["seg_logit_near_zero"][i]) + 2 if dtype == torch.float32 else 0.02 * masks[i].size), (i, bad)
            n_masks += 1
    assert n_masks >= 2
