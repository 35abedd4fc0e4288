#!/usr/bin/env python3
"""Developer tool (GPU box; round 6): randomised scan of the eval head (armnet_mlp_head_f32, fp16 x 2 split with its in-kernel
bf16 x 3 fallback) against a float64 evaluation of the same nn.Sequential — shapes (input width 1..2100, hidden width 1..600,
1..3 hidden layers, batch 1..70 000), input scales from 1e-7 to 1e5 (both fallback thresholds are crossed), weight scales
1e-3..1e3, BatchNorm statistics randomised, some samples with outliers / zeros.  Per case: max |error| of the HIP head, of the
forced bf16 x 3 split and of the fp32 hipBLASLt path, each relative to the magnitude of the terms the logits are summed from
(max |x| * max |w| ... propagated by the float64 absolute-value network).  Prints the worst cases and a summary.
    python tools/head_scan.py [--cases 300] [--seed 0]"""
import argparse
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from armnet_hip.modules import _MLP  # noqa: E402

DEV = "cuda:0"


def abs_network(m, x):
    """float64: the same head with |W|, |b|, |BN scale| and no cancellation — the magnitude of the terms behind every logit"""
    mods = list(m.mlp)
    h = x.double().abs()
    i = 0
    while i < len(mods):
        lin = mods[i]
        h = h @ lin.weight.double().abs().t() + lin.bias.double().abs()
        if i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.BatchNorm1d):
            bn = mods[i + 1]
            s = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)).abs()
            h = h * s + (bn.bias.double() - bn.running_mean.double() * bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)).abs()
            i += 4
        else:
            i += 1
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(a.seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    worst = []
    n_bad = 0
    for c in range(a.cases):
        K0 = [ri(1, 64), ri(1, 2100), 16 * ri(1, 128)][ri(0, 2)]
        nhid = [ri(1, 40), ri(1, 600), 32 * ri(1, 8)][ri(0, 2)]
        nlayers = ri(1, 3)
        B = [ri(1, 70), ri(1, 5000), ri(30000, 70000)][min(2, ri(0, 5) // 2)] if K0 * nhid < 300000 else ri(1, 3000)
        xs = 10.0 ** (ri(-70, 50) / 10.0)
        ws = 10.0 ** (ri(-30, 30) / 10.0)
        torch.manual_seed(1000 + c)
        m = _MLP(K0, nlayers, nhid, 0.0)
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm1d):
                    mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.3)
                    mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) * 1.5 + 0.25)
                    mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                    mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.2)
            m.mlp[0].weight.mul_(ws)
        m = m.eval().to(DEV)
        x = (torch.rand(B, K0, generator=g) * 3.0 - 1.0) * xs
        kind = ri(0, 5)
        if kind == 0 and B > 2:
            x[ri(0, B - 1), ri(0, K0 - 1)] = 7000.0 * max(1.0, xs)          # one outlier: its block redoes in bf16x3
        elif kind == 1:
            x[::3] = 0.0                                                     # dead samples
        elif kind == 2:
            x = x * torch.logspace(-4, 0, K0).unsqueeze(0)                   # columns of very different magnitude
        x = x.to(DEV)
        with torch.no_grad():
            want = copy.deepcopy(m.mlp).double()(x.double())
            mag = abs_network(m, x).clamp_min(1e-300)
            got = m(x).double()
            m.bf16x3 = True
            got3 = m(x).double()
            m.bf16x3 = False
            m.hip_head = False
            blas = m(x).double()
        e = float(((got - want).abs() / mag).max())
        e3 = float(((got3 - want).abs() / mag).max())
        eb = float(((blas - want).abs() / mag).max())
        bad = e > 4.0 * max(eb, e3, 6e-8)
        n_bad += bad
        worst.append((e, e3, eb, K0, nlayers, nhid, B, xs, ws, kind, bad))
    worst.sort(reverse=True)
    print(f"{a.cases} cases (seed {a.seed}); error = max |got - float64| / (magnitude of the terms of that logit)")
    print(f"worst fp16x2-default {worst[0][0]:.2e}; median {sorted(w[0] for w in worst)[len(worst) // 2]:.2e}; "
          f"worst bf16x3 {max(w[1] for w in worst):.2e}; worst hipBLASLt fp32 {max(w[2] for w in worst):.2e}; "
          f"cases where the default is > 4x both others and > 2.4e-7: {n_bad}")
    for w in worst[:12]:
        print(f"  default {w[0]:.2e}  bf16x3 {w[1]:.2e}  blas {w[2]:.2e}   K0={w[3]} nlayers={w[4]} nhid={w[5]} B={w[6]} x-scale {w[7]:.1e} w-scale {w[8]:.1e} kind {w[9]}"
              + ("   <-- " if w[10] else ""))
    sys.exit(1 if n_bad else 0)


if __name__ == "__main__":
    main()
