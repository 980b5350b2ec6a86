"""Per fusion switch, in one process (the switches are module globals of mi355.graph, read when a plan is built): one fp32 training
step of a segmenter with the switch off against all on — ReLU / max-pool decisions that differ between the two forwards (an
activation within an ulp of zero falls on the other side when a dot product is summed with or without fused multiply-adds) and
the parameter gradients that differ most.   python tests/diag/diag_fusion_switches.py [AttentionUNet|R2AttU_Net] [seed]"""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [R, os.path.join(R, "medical-image-segmentation-and-classification_amd"), os.path.join(R, "tests")]
import numpy as np
import torch
from oracle import nets, train as otrain
from mi355 import graph, nn as mnn
from utils.helpers import get_seg_model
from gpu_util import gpu_kinks

name = sys.argv[1] if len(sys.argv) > 1 else "AttentionUNet"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
SW = ("FUSE_POOL", "FUSE_POOL_BWD", "FUSE_GATE_BWD", "FUSE_HEAD", "FUSE_RESIDUAL")


def step(**off):
    for s in SW:
        setattr(graph, s, s not in off)
    graph.Builder.fuse_residual = graph.FUSE_RESIDUAL
    x, y = otrain.synthetic_batch(3, 64, seed=5)
    m = get_seg_model({"AttentionUNet": "attentionunet", "R2AttU_Net": "r2attunet"}[name])
    m.load_state_dict(nets.default_init_state(name, seed=seed))
    m.compute_dtype = torch.float32
    m = m.to("cuda:0").train()
    out = m(x.to("cuda:0"))
    mnn.BCEWithLogitsLoss()(out, y.to("cuda:0")).backward()
    torch.cuda.synchronize()
    relu, pool = gpu_kinks(out._mi355_plan)
    return out.detach().cpu(), {k: p.grad.double().cpu() for k, p in m.named_parameters()}, relu, pool


o_on, g_on, r_on, p_on = step()
for s in SW:
    o, g, r, p = step(**{s: 1})
    flips = sum(int((a != b).sum()) for a, b in zip(r_on, r)) + sum(int((a != b).sum()) for a, b in zip(p_on, p))
    rows = sorted(((float((g_on[k] - g[k]).abs().max() / max(float(g[k].abs().max()), 1e-30)), k, float(g[k].abs().max()),
                    float((g_on[k] - g[k]).norm() / max(float(g[k].norm()), 1e-30))) for k in g), reverse=True)
    print(f"{s}: decisions that differ {flips}; logits {float((o_on - o).abs().max() / o.abs().max()):.2e}")
    for r_ in rows[:6]:
        print("   max-rel %.3e %-40s max|g| %.3e  l2-rel %.3e" % r_)
