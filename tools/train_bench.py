#!/usr/bin/env python3
"""Training-step timing (forward + backward + Adam) of the HIP ARM-Net module on synthetic Criteo-shaped data."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from models.armnet_1h import ARMNetModel

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 1.7
    F, E, H, nfeat = 39, 16, 32, 1_000_000
    torch.manual_seed(0)
    m = ARMNetModel(F, nfeat, E, alpha, H, E, 2, 256, 0.0, False, 2, 256).cuda().train()
    m.check_ids = False
    if os.environ.get("ARMNET_HEAD_GEMM", "mfma") == "hipblaslt":   # A/B: the head's Linear forward / dX on hipBLASLt fp32 (rounds 1-4)
        m.mlp.mfma_train = False
    print("head GEMMs:", "armnet_linear_bf16x3_f32 (forward, dX) + split-K dW" if m.mlp.mfma_train else "hipBLASLt fp32")
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    ids = torch.randint(0, nfeat, (B, F)).cuda(); vals = torch.rand(B, F).cuda(); y = (torch.rand(B) > 0.5).float().cuda()
    lossf = torch.nn.BCEWithLogitsLoss()
    def step():
        loss = lossf(m({"id": ids, "value": vals}), y)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < 0.2:                     # warm-up that also lets the device clocks settle
        step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    # forward+backward of the block alone
    z = None
    def fb():
        out = m({"id": ids, "value": vals}); out.sum().backward()
    for _ in range(2): fb()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fb()
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / n
    print(f"B={B} alpha={alpha}: train step {dt*1e3:.2f} ms ({B/dt/1e3:.0f} k samples/s); fwd+bwd only {dt2*1e3:.2f} ms")
    # the same step replayed as one hipGraph
    from armnet_hip.modules import GraphedTrainStep
    opt2 = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
    gstep = GraphedTrainStep(m, opt2, lossf, ids, vals, y)
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < 0.2:
        gstep(ids, vals, y); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): gstep(ids, vals, y)
    torch.cuda.synchronize(); dt3 = (time.perf_counter() - t0) / n
    print(f"B={B} alpha={alpha}: graphed train step {dt3*1e3:.2f} ms ({B/dt3/1e3:.0f} k samples/s)")

main()
