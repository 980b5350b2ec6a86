"""CPU: the oracle's block functions vs the reference blocks' stored results (blocks.npz)."""
import numpy as np
import pytest
import torch

import block_cases as bc
from oracle import nets

ORACLE = {
    "basic_block": lambda sd, ins: nets.double_conv(sd, "", ins[0], True),
    "UpConv": lambda sd, ins: nets.up_conv(sd, "", ins[0], True),
    "AttentionGate": lambda sd, ins: nets.attention_gate(sd, "", ins[0], ins[1], True),
    "Recurrent_block": lambda sd, ins: nets.recurrent(sd, "", ins[0], 5, True),
    "RRCNN_block": lambda sd, ins: nets.rrcnn(sd, "", ins[0], 2, True),
    "BasicBlock_s2": lambda sd, ins: nets._basic_block(sd, "", ins[0], 2, True),
    "DecoderBlock": lambda sd, ins: nets._decoder_block(sd, "", ins[0], ins[1], True),
}
SHAPES = {   # state_dict layouts of the reference blocks, via the oracle's spec helpers
    "basic_block": lambda s: nets._spec_double_conv(s, "", 32, 64),
    "UpConv": lambda s: nets._spec_up(s, "", 64, 32),
    "AttentionGate": lambda s: nets._spec_gate(s, "", 64, 32),
    "Recurrent_block": lambda s: (s.conv(".conv.0", 32, 32, 3), s.bn(".conv.1", 32)),
    "RRCNN_block": lambda s: nets._spec_rrcnn(s, "", 32, 64),
    "BasicBlock_s2": lambda s: (s.conv(".conv1", 32, 64, 3, False), s.conv(".conv2", 64, 64, 3, False), s.bn(".bn1", 64),
                                s.bn(".bn2", 64), s.conv(".identity.0", 32, 64, 1, False), s.bn(".identity.1", 64)),
    "DecoderBlock": lambda s: (nets._spec_double_conv(s, ".basic_block", 96, 32), s.convT(".up_sample", 64, 64, 2)),   # registration order of ResnetUnet.py:20-21
}


def block_state(tag):
    """Closed-form state dict keyed like the oracle helpers expect (leading '.')."""
    sp = nets._Spec()
    SHAPES[tag](sp)
    shapes = {}
    for k, (shape, kind, _) in sp.entries.items():
        shapes[k] = torch.zeros(shape, dtype=torch.int64 if kind == "nbt" else torch.float32)
        if kind == "rv":
            shapes[k] += 1
    filled = bc.fill(tag, shapes)
    # fresh BN buffers (the reference modules were constructed fresh, then loaded with the same fill)
    return filled


@pytest.mark.parametrize("tag", bc.ORDER)
def test_block_oracle_matches_reference(tag):
    z = bc.load()
    sd = block_state(tag)
    ins = [t.clone().requires_grad_(True) for t in bc.inputs(tag)]
    pk = nets.param_keys(sd)
    for k in pk:
        sd[k].requires_grad_(True)
    out = ORACLE[tag](sd, ins)
    ref = z[tag + "/out"]
    assert np.abs(out.detach().numpy() - ref).max() < 2e-5 * np.abs(ref).max()
    (out * bc.out_weight(tag, out.shape)).sum().backward()
    for i, t in enumerate(ins):
        r = z[f"{tag}/din{i}"]
        assert np.abs(t.grad.numpy() - r).max() < 1e-4 * np.abs(r).max()
    for k in pk:
        got, r, nrm = bc.expected_grad(z, tag, k.lstrip("."), sd[k].grad.detach())
        assert np.abs(got - r).max() < 1e-4 * (np.abs(r).max() + 1e-3 * nrm + 1e-9)
    for k, v in sd.items():
        if nets.is_buffer(k):
            r = z[f"{tag}/buf/{k.lstrip('.')}"]
            assert np.abs(v.detach().numpy().astype(np.float64) - r).max() < 1e-5 * (np.abs(r).max() + 1e-6)
