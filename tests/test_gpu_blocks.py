"""-m gpu: the drop-in building blocks vs the REFERENCE blocks' stored results (blocks.npz):
output, input gradients, parameter gradients and BN running statistics, fp32 mode.
Blocks are shallow, so fp32 agrees to ~1e-6; the bound asserted is 1e-4 relative-to-max
(north-star bound 1e-3).  bf16 mode is checked at 3e-2 on outputs."""
import numpy as np
import pytest
import torch

import block_cases as bc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make(tag):
    from mi355.engine import Net
    from models.segmentation_models import _blocks as B
    from models.classification_models.ResNet import BasicBlock
    args = bc.CASES[tag][0]
    if tag == "basic_block":
        blk, low = B.conv_bn_relu_x2(*args), lambda g, b, xs: g.seq(b, xs[0])
    elif tag == "UpConv":
        blk, low = B.UpConv(*args), lambda g, b, xs: g.seq(b.up, xs[0])
    elif tag == "AttentionGate":
        blk, low = B.AttentionGate(*args), lambda g, b, xs: g.gate(b, g=xs[0], x=xs[1])
    elif tag == "Recurrent_block":
        blk, low = B.Recurrent_block(args[0], args[1], t=args[2]), lambda g, b, xs: b.lower(g, xs[0])
    elif tag == "RRCNN_block":
        blk, low = B.RRCNN_block(args[0], args[1], t=args[2]), lambda g, b, xs: b.lower(g, xs[0])
    elif tag == "DecoderBlock":
        from models.segmentation_models.ResnetUnet import DecoderBlock
        # the block net has ONE input: `down` (4 x 4) rides in as its nearest x2 up-sampling next to `skip` (8 x 8) and is recovered
        # exactly by a 2 x 2 max-pool of four equal values; its gradient lands on the first element of every 2 x 2 block
        blk, low = DecoderBlock(*args), lambda g, b, xs: b.lower(g, g.maxpool(xs[0], 2, 2, 0), xs[1])
    else:
        blk, low = BasicBlock(args[0], args[1], stride=args[2]), lambda g, b, xs: b.lower(g, xs[0])
    chans = [s[1] for s in bc.CASES[tag][1]]

    class BlockNet(Net):
        def __init__(self):
            super().__init__()
            self.block = blk

        def build(self, g, x):
            g.want_input_grad(x)
            xs, o = [], 0
            for c in chans:
                xs.append(g.slice_channels(x, o, c) if len(chans) > 1 else x)
                o += c
            g.tensor_output(low(g, self.block, xs))

    blk.load_state_dict(bc.fill(tag, blk.state_dict()))
    return BlockNet()


def _net_input(tag):
    ins = bc.inputs(tag)
    if tag == "DecoderBlock":
        ins = [ins[0].repeat_interleave(2, 2).repeat_interleave(2, 3), ins[1]]
    return torch.cat(ins, 1)


@pytest.mark.parametrize("tag", bc.ORDER)
def test_block_fp32_matches_reference(tag):
    z = bc.load()
    net = _make(tag)
    net.compute_dtype = torch.float32
    net = net.to(DEV).train()
    ins = bc.inputs(tag)
    x = _net_input(tag).to(DEV)
    out = net(x)
    ref = z[tag + "/out"]
    w = bc.out_weight(tag, ref.shape)
    (out * w.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    tol = 1e-4
    assert np.abs(out.detach().cpu().numpy() - ref).max() < tol * np.abs(ref).max()
    din = out._mi355_plan.input_grad.view(x.shape).cpu()
    o = 0
    for i, t in enumerate(ins):
        r = z[f"{tag}/din{i}"]
        got = din[:, o:o + t.shape[1]]
        if got.shape[2:] != t.shape[2:]:            # DecoderBlock's `down`: fold the 2 x 2 blocks of its up-sampled carrier
            got = got.reshape(t.shape[0], t.shape[1], t.shape[2], 2, t.shape[3], 2).sum((3, 5))
        got = got.numpy()
        o += t.shape[1]
        assert np.abs(got - r).max() < tol * np.abs(r).max(), (i, np.abs(got - r).max(), np.abs(r).max())
    # conv biases in front of a train-mode BN have a mathematically zero gradient (pure round-off in
    # both implementations), hence the block-wide absolute floor
    gscale = max(float(np.abs(z[f]).max()) for f in z.files if f.startswith(tag + "/grad/"))
    for k, p in net.block.named_parameters():
        got, r, nrm = bc.expected_grad(z, tag, k, p.grad.detach().cpu())
        assert np.abs(got - r).max() < tol * (np.abs(r).max() + gscale), k
    for k, v in net.block.state_dict().items():
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            r = z[f"{tag}/buf/{k}"]
            assert np.abs(v.cpu().numpy().astype(np.float64) - r).max() < 1e-5 * (np.abs(r).max() + 1e-6), k


@pytest.mark.parametrize("tag", bc.ORDER)
def test_block_bf16_close_to_reference(tag):
    z = bc.load()
    net = _make(tag)
    net.compute_dtype = torch.bfloat16
    net = net.to(DEV).train()
    x = _net_input(tag).to(DEV)
    with torch.no_grad():
        out = net(x)
    ref = z[tag + "/out"]
    lim = 6e-2 if tag in ("Recurrent_block", "RRCNN_block") else 3e-2     # 6 / 12 stacked bf16 conv+BN
    assert np.abs(out.cpu().numpy() - ref).max() < lim * np.abs(ref).max()


@pytest.mark.parametrize("t", [2, 7, 10])
def test_recurrent_block_deferred_gradient_sum_any_number_of_applications(t):
    """Recurrent_block (R2AttU_Net.py:29-45) applied t + 1 times: the block input's gradient is the sum of t incoming gradients plus
    the first convolution's data gradient.  With DEFER_POST the BatchNorm apply passes leave them pending and
    mi355_bn_bwd_apply_post4 adds up to four at a time (t = 7: a flush in the middle and one at the end; t = 10: two in the middle),
    without it every pass read-modify-writes the gradient.  Same terms, another order: fp32 results agree to summation-order
    rounding; the forward is untouched (bit-identical)."""
    from mi355 import graph
    from mi355.engine import Net
    from models.segmentation_models import _blocks as B
    args = bc.CASES["Recurrent_block"][0]
    x = _net_input("Recurrent_block").to(DEV)
    res = {}
    saved = graph.DEFER_POST
    try:
        for defer in (True, False):
            graph.DEFER_POST = defer
            blk = B.Recurrent_block(args[0], args[1], t=t)
            blk.load_state_dict(bc.fill("Recurrent_block", blk.state_dict()))

            class BlockNet(Net):
                def __init__(self):
                    super().__init__()
                    self.block = blk

                def build(self, g, xx):
                    g.want_input_grad(xx)
                    g.tensor_output(self.block.lower(g, xx))

            net = BlockNet()
            net.compute_dtype = torch.float32
            net = net.to(DEV).train()
            out = net(x)
            w = bc.out_weight("Recurrent_block", tuple(out.shape))
            (out * w.to(DEV)).sum().backward()
            torch.cuda.synchronize()
            plan = out._mi355_plan
            names = [l.name for l in plan.bwd]
            res[defer] = (out.detach().cpu(), plan.input_grad.view(x.shape).cpu().clone(),
                          {k: p.grad.detach().cpu().clone() for k, p in net.block.named_parameters()}, names)
    finally:
        graph.DEFER_POST = saved
    on, off = res[True], res[False]
    want, pend = 0, 0                      # Builder._bn_bwd's rule, replayed: t applications add the block input
    for i in range(1, t + 1):
        if i < t and pend < 4:
            pend += 1
        else:
            want, pend = want + (pend > 0), 0
    assert on[3].count("mi355_bn_bwd_apply_post4") == want and want >= 1
    assert "mi355_bn_bwd_apply_post4" not in off[3]
    assert torch.equal(on[0], off[0])
    assert float((on[1] - off[1]).abs().max()) <= 2e-5 * float(off[1].abs().max())
    gmax = max(float(g.abs().max()) for g in off[2].values())
    for k, g in off[2].items():
        assert float((on[2][k] - g).abs().max()) <= 2e-5 * (float(g.abs().max()) + 1e-3 * gmax), k
