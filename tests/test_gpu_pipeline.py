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


def _he(sd):
    """default init has gain 1/sqrt(3); eval-mode BN with fresh running statistics is the identity, so rescale the
    convolutions to He gain to keep activations O(1) through the depth"""
    for v in sd.values():
        if v.dim() == 4:
            v.mul_(6 ** 0.5)
    return sd


def _fixture():
    cls_sd = _he(nets.default_init_state("ResNet18", seed=3, num_classes=3, head_dropout=True))
    seg_sd = _he(nets.default_init_state("AttentionUNet", seed=4))
    x = torch.randn(16, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():        # centre the logits over the batch: the three classes all occur
        cls_sd["fc.1.bias"] = cls_sd["fc.1.bias"] - nets.resnet18({k: v.clone() for k, v in cls_sd.items()}, x, False).mean(0)
        z = nets.resnet18({k: v.clone() for k, v in cls_sd.items()}, x, False)
    top = z.sort(1, descending=True).values
    return cls_sd, seg_sd, x, top[:, 0] - top[:, 1]


def _models(dtype, cls_sd, seg_sd):
    from models.classification_models.ResNet import ResNet18
    from models.segmentation_models.AttentionUNet import AttentionUNet
    from utils.helpers import add_dropout_to_fc
    cm = ResNet18(num_classes=3)
    add_dropout_to_fc(cm)
    cm.load_state_dict(cls_sd)
    sm = AttentionUNet()
    sm.load_state_dict(seg_sd)
    cm.compute_dtype = sm.compute_dtype = dtype
    return cm, sm


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_joint_pipeline_matches_per_image_oracle(dtype):
    from utils.pipeline import JointPipeline
    cls_sd, seg_sd, x, margin = _fixture()
    cm, sm = _models(dtype, cls_sd, seg_sd)
    ref = opipe.process_batch("ResNet18", cls_sd, "AttentionUNet", seg_sd, x)
    n_pos = sum(r[2] is not None for r in ref)
    assert 2 <= n_pos <= len(ref) - 2, "degenerate fixture: the batch must mix COVID and non-COVID predictions"
    pipe = JointPipeline(cm, sm, device=DEV, bucket=4)
    got = pipe.process_batch(x)
    sure = margin > (0.0 if dtype == torch.float32 else 1.0)
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
