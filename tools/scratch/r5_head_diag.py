#!/usr/bin/env python3
"""round 5 diagnostic: gradients of one ARM-Net training step at B = 4096 with the head's GEMMs on hipBLASLt (twice: run-to-run
spread of the float atomics) and on armnet_linear_bf16x3_f32, per parameter, relative to the parameter's largest gradient"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from models.armnet_1h import ARMNetModel
DEV = "cuda:0"
g = torch.Generator().manual_seed(2)
F, E, H, nfeat, B = 39, 16, 32, 5000, 4096
ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
vals = torch.rand(B, F, generator=g).to(DEV)
y = (torch.rand(B, generator=g) > 0.5).float().to(DEV)
res = []
def build_host():
    torch.manual_seed(11)
    m = ARMNetModel(F, nfeat, E, 1.7, H, E, 2, 256, 0.0, False, 2, 256).train()
    with torch.no_grad():
        m.attn_layer.query.mul_(4.0)
        m.embedding.embedding.weight.normal_(0, 0.5)
    return m


def host_reference():
    m = build_host()
    logits = m({"id": ids.cpu(), "value": vals.cpu().clone()})
    torch.nn.BCEWithLogitsLoss()(logits, y.cpu()).backward()
    return logits.detach().to(DEV), {k: p.grad.clone().to(DEV) for k, p in m.named_parameters()}

for mfma in (False, False, False, True, False):
    m = build_host().to(DEV).train()
    m.mlp.mfma_train = mfma
    keep = {}
    x_arm_hook = []
    logits = m({"id": ids, "value": vals.clone()})
    torch.nn.BCEWithLogitsLoss()(logits, y).backward()
    res.append((logits.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}))
res.append(host_reference())
for name, (a, b) in (("host ATen chain vs run 0 (hipBLASLt, first in the process)", (5, 0)), ("host vs run 1 (hipBLASLt)", (5, 1)),
                     ("host vs run 2 (hipBLASLt)", (5, 2)), ("host vs run 3 (matrix-core head)", (5, 3)), ("host vs run 4 (hipBLASLt)", (5, 4)),
                     ("run 0 vs run 1", (0, 1)), ("run 1 vs run 2", (1, 2))):
    print(name, "logits", float((res[a][0] - res[b][0]).abs().max()))
    for k, gref in res[a][1].items():
        d = (res[b][1][k] - gref).abs()
        print(f"   {k:40s} max|g| {float(gref.abs().max()):.3e}  max diff {float(d.max()):.3e}  rel {float(d.max()) / float(gref.abs().max() + 1e-30):.2e}")
