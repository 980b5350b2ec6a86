"""-m gpu: config C5's classifier in its own precision.  ``vgg16_bn`` (the torchvision layout helpers.py:158-166 loads) trained
with the reference's classification step — CrossEntropy(label_smoothing 0.1), GradScaler, unscale, clip 1.0, AdamW(wd 5e-4)
(helpers.py:245, 285, 320-336) — in fp16 on the HIP path for 12 steps, next to the fp32 CPU oracle from identical weights and
batches.  Bound: arg-max of the held-out logits identical, final loss within 2 %.  The task (class = colour cast)
is learnable in 12 steps and the oracle ends with top-2 margins > 1, so the comparison is not decided by rounding."""
import pytest
import torch

from oracle import nets
from oracle import train as otrain

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _task(b, hw, seed):
    g = torch.Generator().manual_seed(seed)
    y = torch.randint(0, 3, (b,), generator=g)
    x = 0.7 * torch.randn(b, 3, hw, hw, generator=g)
    x[torch.arange(b), y] += 1.0
    return x, y


@pytest.mark.parametrize("dtype", [torch.float16])          # C5's own precision (fp32 runs the same code path: 3 % after 12 steps)
def test_vgg16_bn_training_matches_oracle(dtype):
    from mi355 import amp as mamp, nn as mnn, optim as moptim
    from utils.helpers import get_class_model
    hw, b, steps, lr = 64, 8, 12, 2e-5
    batches = [_task(b, hw, s) for s in range(4)]
    xv, yv = _task(32, hw, 99)
    sd0 = nets.default_init_state("VGG16_BN", seed=0, num_classes=3, head_dropout=True)

    sd = {k: v.clone() for k, v in sd0.items()}
    opt = otrain.AdamW(nets.param_keys(sd), lr)
    for i in range(steps):
        ref_loss, _, _ = otrain.train_step("VGG16_BN", sd, *batches[i % 4], opt, False)
    with torch.no_grad():
        zr = nets.vgg16_bn({k: v.clone() for k, v in sd.items()}, xv, True)
    top = zr.sort(1, descending=True).values
    assert float((zr.argmax(1) == yv).float().mean()) == 1.0 and float((top[:, 0] - top[:, 1]).min()) > 1.0

    m, head = get_class_model("vgg16_bn")
    assert head == "classifier"
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0                                   # dropout streams cannot be matched across implementations
    m.load_state_dict(sd0)
    m.compute_dtype = dtype
    m = m.to(DEV).train()
    o = moptim.AdamW(m.parameters(), lr=lr, weight_decay=5e-4)
    crit = mnn.CrossEntropyLoss(label_smoothing=0.1)
    scaler = mamp.GradScaler(enabled=dtype == torch.float16)          # torch defaults: 65536, x2 / x0.5, interval 2000
    done = it = 0
    while done < steps and it < steps + 8:
        x, y = batches[done % 4]
        o.zero_grad(set_to_none=True)
        loss = crit(m(x.to(DEV)), y.to(DEV))
        scaler.scale(loss).backward()
        scaler.unscale_(o)
        moptim.clip_grad_norm_(m.parameters(), 1.0)
        scaler.step(o)
        scaler.update()
        it += 1
        done = int(o._st[0]["step"])
    assert done == steps
    with torch.no_grad():
        z = m(xv.to(DEV)).float().cpu()
    assert torch.equal(z.argmax(1), zr.argmax(1))
    assert abs(float(loss.detach()) - ref_loss) <= 0.02 * ref_loss, (float(loss.detach()), ref_loss)
    assert float((z - zr).abs().max()) <= (0.06 if dtype == torch.float32 else 0.3) * float(zr.abs().max())     # (12 optimisation steps apart)
