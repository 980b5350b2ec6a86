"""Shared description of the block-level fixtures in tests/golden/blocks.npz (generated from the
reference's block classes by oracle/make_golden.py; weights/inputs are closed forms, so only the
expected results are stored)."""
import os

import numpy as np
import torch

from oracle import nets

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "blocks.npz")

CASES = {   # tag -> (constructor args, input shapes) — keep in sync with oracle/make_golden.py:BLOCK_CASES
    "basic_block": ((32, 64), [(2, 32, 12, 12)]),
    "UpConv": ((64, 32), [(2, 64, 6, 6)]),
    "AttentionGate": ((64, 64, 32), [(2, 64, 8, 8), (2, 64, 8, 8)]),
    "Recurrent_block": ((32, 32, 5), [(2, 32, 8, 8)]),
    "RRCNN_block": ((32, 64, 2), [(2, 32, 8, 8)]),
    "BasicBlock_s2": ((32, 64, 2), [(2, 32, 8, 8)]),
    "DecoderBlock": ((96, 32), [(2, 64, 4, 4), (2, 32, 8, 8)]),      # ResnetUnet.py:17-27 (imported behind empty torchvision stand-ins)
}
ORDER = list(CASES)


def inputs(tag):
    ti = ORDER.index(tag)
    return [nets.closed_form_tensor(s, 100.0 + 10 * ti + j) for j, s in enumerate(CASES[tag][1])]


def out_weight(tag, shape):
    return nets.closed_form_tensor(tuple(shape), 200.0 + ORDER.index(tag))


def fill(tag, state_dict):
    return nets.closed_form_fill(state_dict, salt=10.0 * ORDER.index(tag))


def load():
    return np.load(GOLDEN, allow_pickle=False)


def expected_grad(z, tag, key, got):
    """Compare a parameter gradient with the stored one (large tensors keep every 7th element)."""
    ref = z[f"{tag}/grad/{key}"]
    g = got.reshape(-1)
    g = g if g.numel() <= 4096 else g[::7]
    return g.numpy(), ref, float(z[f"{tag}/gradnorm/{key}"])
