#!/usr/bin/env python3
"""Developer tool (CPU, numpy; round 5, verdict item 1): wave-level replay of the generic-alpha sparse-map solver of the
matrix-core forward on the gates of bench.py's two weight regimes, ONE-sample wave groups (16 rows per pass, a row
spread over 4 lanes x 10 elements = 5 pairs per lane), counting per 16-neuron pass
  * wave-uniform evaluations (the loop runs to the slowest row), split into CHEAP ones (no transcendental: p = t^2)
    and GENERIC ones (p = t^r through exp2 / log2: 4 transcendentals per pair and lane),
  * transcendental issues per lane = sum over generic evaluations of 4 x (pairs the wave evaluates).
Variants:
  base      Newton from the left from tau0 = max(mx - 1, mean - d^-(alpha-1))         (the product kernel of round 4)
  warm2     t^r >= t^2 on [0, 1] for r <= 2, so the root of sum t^2 = 1 is a LOWER bound of the root of sum t^r = 1:
            cheap Newton on sum t^2 first (until every row's residual is below `theta`, at most `kmax` evaluations), then
            the generic Newton from there
  compact   after generic evaluation `c_after`, dead elements (x <= tau: dead for good, Newton from the left is monotone)
            are dropped and the row's live elements re-spread over its 4 lanes: the wave evaluates
            ceil(max_rows(live) / 8) pairs per lane from then on
  lin       the last evaluation is replaced by a first-order update p - r u dtau when the Newton step is below `eps_lin`
            (BUILT: fused_mfma_kernel.h, SparseMapCfg.lin_tol)
  halley    a Halley step with f'' estimated from the last two slopes (free), overshoots repaired by Newton from the right
    python tools/solver_sim_r5.py [--alphas 1.7 1.5 2.0] [--batch 2048]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'arm-net_amd')):
    sys.path.insert(0, _p)
import bench  # noqa: E402


def gates_for(regime, batch):
    a = argparse.Namespace(gpus=1, steps=1, warmup=0, alpha=2.0, regime=regime, batch=batch, nfield=39, nfeat=1_000_000,
                           nemb=16, nhid=32, nhead=1, ids="uniform", shard="replicate", micro_batches=1, rotate=1,
                           ensemble=False)
    m = bench.build_model(a, torch.device("cpu"), regime=regime)
    ids, vals, _, _ = bench.make_batch(a, 0, torch.device("cpu"), 0)
    with torch.no_grad():
        v = vals.clamp(1e-3, 1.0)
        x = m.embedding.embedding.weight[ids] * v.unsqueeze(2)
        W = m.attn_layer.bilinear_w.weight
        q = m.attn_layer.query
        qf = (q @ W) * (a.nemb ** -0.5)
        g = torch.einsum('bfe,oe->bof', x, qf)
    g = g.numpy().astype(np.float32)
    B, O, F = g.shape
    return g.reshape(B, O // 16, 16, F).reshape(-1, 16, F)      # [wave-passes, 16 rows, F]


def solve(X, alpha, warm=None, theta=0.1, kmax=8, compact_after=None, eps_lin=0.0, tol=6e-7, tau_tol=2e-7, maxit=40):
    """returns dict of per-wave-pass means"""
    am1 = np.float32(alpha - 1)
    r = np.float32(1.0) / am1
    Xs = (X * am1).astype(np.float32)
    W, R, F = Xs.shape
    mx = Xs.max(-1)
    mean = Xs.mean(-1, dtype=np.float32)
    tau = np.maximum(mx - 1, mean - np.float32((1.0 / F) ** (alpha - 1))).astype(np.float32)
    cheap = np.zeros(W, int)
    # ---- cheap phase: Newton on sum t^2 = 1 (root <= the generic root for r <= 2) --------------------------
    if warm == "t2" and r <= 2.0:
        active = np.ones((W, R), bool)
        for it in range(kmax):
            wa = active.any(1)
            if not wa.any():
                break
            cheap += wa
            t = np.clip(Xs - tau[..., None], 0, None)
            S = (t * t).sum(-1)
            D = 2 * t.sum(-1)
            f = S - 1
            tn = tau + f / np.maximum(D, 1e-30)
            act = (f > theta) & (tn > tau) & active
            # rows below theta still take their (safe) step: it costs nothing and moves them right
            take = (f > 0) & (tn > tau) & active
            tau = np.where(take, tn, tau).astype(np.float32)
            active = act
    # ---- generic phase ------------------------------------------------------------------------------------------
    active = np.ones((W, R), bool)
    gen = np.zeros(W, int)
    trans = np.zeros(W, float)            # transcendental issues per lane
    pairs_now = np.full(W, 5.0)
    if compact_after == 0:                # compaction right behind the cheap phase
        live = (Xs > tau[..., None]).sum(-1)
        pairs_now = np.maximum(np.ceil(live.max(1) / 8.0), 1.0)
    for it in range(maxit):
        wa = active.any(1)
        if not wa.any():
            break
        gen += wa
        trans += wa * 4 * pairs_now
        t = np.clip(Xs - tau[..., None], 0, None)
        with np.errstate(divide='ignore', invalid='ignore'):
            u = np.where(t > 0, np.exp2((r - 1) * np.log2(t, where=t > 0, out=np.zeros_like(t))), 0).astype(np.float32)
        S = (u * t).sum(-1)
        D = r * u.sum(-1)
        f = S - 1
        step = f / np.maximum(D, 1e-30)
        tn = tau + step
        thr = np.maximum(tol, tau_tol * min(1.0, float(am1) / 0.7) * D)
        act = (f > thr) & (tn > tau) & active
        if eps_lin > 0:
            lin = act & (step < eps_lin)          # first-order update instead of one more evaluation
            act &= ~lin
        tau = np.where(act, tn, tau).astype(np.float32)
        active = act
        if compact_after is not None and it + 1 == compact_after:
            live = (Xs > tau[..., None]).sum(-1)                     # per row
            pairs_now = np.ceil(np.where(active, live, 0).max(1) / 8.0)      # rows still active set the trip count
            pairs_now = np.maximum(pairs_now, 1.0)
    return dict(cheap=cheap.mean(), gen=gen.mean(), trans=trans.mean(), tau=tau)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--alphas", type=float, nargs="*", default=[1.7, 1.5, 1.3])
    ap.add_argument("--batch", type=int, default=2048)
    a = ap.parse_args()
    # issue-slot model per evaluation and wave (alpha = 2 evaluation ~ 7 us, alpha = 1.7 ~ 20 us per batch):
    #   overhead (LDS reduce, step, ballots) 25; cheap pair 4; generic pair 22 (4 quarter-rate transcendentals = 16)
    def cost(res, compaction=False):
        return res["cheap"] * (25 + 5 * 4) + res["gen"] * 25 + res["trans"] / 4 * 22 + (80 if compaction else 0)
    for regime in ("fresh", "stress"):
        X = gates_for(regime, a.batch)
        for alpha in a.alphas:
            base = solve(X, alpha)
            print(f"{regime:6s} alpha {alpha}: base  cheap {base['cheap']:.2f} generic {base['gen']:.2f} "
                  f"trans/lane {base['trans']:.1f} slots {cost(base):.0f}")
            for name, kw, comp in (
                    ("compact after 1", dict(compact_after=1), True),
                    ("compact after 2", dict(compact_after=2), True),
                    ("warm t2 theta .3", dict(warm="t2", theta=0.3), False),
                    ("warm t2 theta .1", dict(warm="t2", theta=0.1), False),
                    ("warm t2 theta .03", dict(warm="t2", theta=0.03), False),
                    ("warm t2 theta .01", dict(warm="t2", theta=0.01), False),
                    ("warm t2 kmax 3", dict(warm="t2", theta=0.0, kmax=3), False),
                    ("warm t2 kmax 4", dict(warm="t2", theta=0.0, kmax=4), False),
                    ("warm .1 + lin 1e-4", dict(warm="t2", theta=0.1, eps_lin=1e-4), False),
                    ("warm .1 + lin 3e-5", dict(warm="t2", theta=0.1, eps_lin=3e-5), False),
                    ("warm .1 + compact 1", dict(warm="t2", theta=0.1, compact_after=1), True),
                    ("warm .1 + compact 0", dict(warm="t2", theta=0.1, compact_after=0), True),
            ):
                res = solve(X, alpha, **kw)
                dt = np.abs(res["tau"] - base["tau"]).max()
                print(f"    {name:22s} cheap {res['cheap']:.2f} generic {res['gen']:.2f} trans/lane {res['trans']:.1f} "
                      f"(x{base['trans'] / max(res['trans'], 1e-9):.2f} fewer) slots {cost(res, comp):.0f} "
                      f"(x{cost(base) / cost(res, comp):.2f})  max|dtau| {dt:.1e}")


def solve2(X, alpha, halley=True, eps_lin=0.0, cap=0.5, tol=6e-7, tau_tol=2e-7, maxit=40, first_norm=False):
    f32=np.float32
    am1 = f32(alpha - 1); r = f32(1.0)/am1
    Xs = (X*am1).astype(f32)
    W,R,F = Xs.shape
    mx = Xs.max(-1); mean = Xs.mean(-1, dtype=f32)
    tau = np.maximum(mx-1, mean - f32((1.0/F)**(alpha-1))).astype(f32)
    active = np.ones((W,R), bool)
    gen = np.zeros(W, int)
    pt = np.zeros_like(tau); pD = np.zeros_like(tau); have = np.zeros((W,R), bool)
    overs = 0
    lin_pending = np.zeros((W,R), bool)
    for it in range(maxit):
        wa = active.any(1)
        if not wa.any(): break
        gen += wa
        t = np.clip(Xs - tau[...,None], 0, None)
        with np.errstate(all='ignore'):
            if alpha == 2.0:
                u = (t>0).astype(f32)
            else:
                u = np.where(t>0, np.exp2((r-1)*np.log2(np.where(t>0,t,1))), 0).astype(f32)
        S = (u*t).sum(-1, dtype=f32); D = (r*u.sum(-1, dtype=f32)).astype(f32)
        f = S - 1
        newt = f/np.maximum(D, f32(1e-30))
        step = newt.copy()
        if halley:
            with np.errstate(all='ignore'):
                f2 = (pD - D)/(tau - pt)            # secant estimate of f'' (>= 0 for convex f)
                den = 1 - f*f2/(2*D*D)
                ok = have & np.isfinite(den) & (f2 > 0) & (f > 0)
                den = np.clip(den, cap, 1.0)
                step = np.where(ok, newt/den, newt)
        if first_norm and it == 0:
            with np.errstate(all='ignore'):
                stepn = (r*S/D)*(1-np.exp2(-np.log2(np.maximum(S,1e-30))/r))
            step = np.where(f>0, np.maximum(step, stepn), step)
        thr = np.maximum(tol, tau_tol*min(1.0, float(am1)/0.7)*D)
        tn = (tau + step).astype(f32)
        conv = ~(np.abs(f) > thr) | (tn == tau)
        act = active & ~conv
        overs += (active & (f < -thr)).sum()
        if eps_lin > 0:
            lin = act & (np.abs(step) < eps_lin)
            act &= ~lin
        else:
            lin = np.zeros_like(act)
        upd = act | lin
        pt = np.where(upd, tau, pt); pD = np.where(upd, D, pD); have |= upd
        tau = np.where(upd, tn, tau).astype(f32)
        active = act
    # residual check in float64 at final tau
    t = np.clip(Xs.astype(np.float64) - tau[...,None].astype(np.float64), 0, None)
    res = (t**float(r)).sum(-1) - 1
    return gen.mean(), np.abs(res).max(), overs, np.bincount(gen)



def support_stats(X, alpha):
    """supports of the exact root (float64 bisection) and of the kernel's start, per row and per wave (max of 16 rows)"""
    am1 = alpha - 1
    r = 1 / am1
    Xs = (X * np.float32(am1)).astype(np.float64)
    F = Xs.shape[-1]
    mx = Xs.max(-1)
    lo, hi = mx - 1, mx.copy()
    for _ in range(60):
        m = (lo + hi) / 2
        fm = (np.clip(Xs - m[..., None], 0, None) ** r).sum(-1) - 1
        lo = np.where(fm > 0, m, lo)
        hi = np.where(fm > 0, hi, m)
    tau0 = np.maximum(mx - 1, Xs.mean(-1) - (1 / F) ** am1)
    s_root, s_start = (Xs > lo[..., None]).sum(-1), (Xs > tau0[..., None]).sum(-1)
    return s_root.mean(), s_root.max(1).mean(), s_start.mean(), s_start.max(1).mean()


def more(batch):
    """the other variants weighed in round 5: support sizes, the first-order finish alone, secant-Halley steps"""
    X = gates_for("stress", batch)
    for alpha in (2.0, 1.7, 1.5):
        a, b, c, d = support_stats(X, alpha)
        print(f"stress alpha {alpha}: support at the root {a:.1f} of {X.shape[-1]} (slowest row of a wave {b:.1f}); at the start {c:.1f} ({d:.1f})")
    for alpha in (1.7, 1.5, 1.3, 1.9):
        base = solve(X, alpha)
        print(f"stress alpha {alpha}: Newton {base['gen']:.2f} evaluations per pass; first-order finish below eps: "
              + ", ".join(f"{eps:g}: {solve(X, alpha, eps_lin=eps)['gen']:.2f}" for eps in (1e-5, 3e-5, 1e-4, 3e-4)))
        for kw in (dict(halley=True), dict(halley=True, eps_lin=1e-4), dict(halley=True, eps_lin=1e-4, first_norm=True)):
            g = solve2(X, alpha, **kw)
            print(f"    secant-Halley {kw}: {g[0]:.2f} evaluations, max |residual| {g[1]:.1e}, overshoots {g[2]}")


if __name__ == "__main__":
    main()
    more(1024)
