#!/usr/bin/env python3
"""Developer tool (CPU, numpy; round 6, round-5 verdict next 6): ONE more replay of the generic-alpha Newton solver of the
matrix-core forward on bench.py's trained-like gates (one-sample wave groups: 16 rows per pass, the loop runs to the slowest
row) — the only direction round 5 left unexplored: cheaper EARLY evaluations.

  approx    while a row's residual is large, t^(r-1) comes from the exponent-field trick
                bits(t^p) ~ p * (bits(t) - B) + B,  B = 0x3f800000 - 0x5c416 * ...   (~3 % relative error, no transcendental:
            v_cvt_f32_u32, v_pk_fma_f32, v_cvt_u32_f32, v_cndmask for t = 0: 7 full-rate instructions per PAIR against
            4 quarter-rate transcendentals + 2 multiplies = 72 issue cycles)
            and the Newton step is made SAFE — Newton from the left must stay left of the root —
                step = (S~ / (1 + eps) - 1) / (D~ / (1 - eps))
            with eps the trick's error bound; a wave switches to exact evaluations once no row of it can take a safe step
            (S~ / (1 + eps) <= 1 + theta) and finishes like the product kernel (first-order finish below lin_tol).
Counts per 16-neuron pass: approximate and exact wave evaluations, issue cycles of the element work
(approximate pair 28, exact pair 72; 5 pairs per lane) and the worst |tau - tau_base|.
    python tools/solver_sim_r6.py [--alphas 1.7 1.3 1.9] [--batch 2048]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from solver_sim_r5 import gates_for  # noqa: E402

f32 = np.float32


def pow_trick(t, p):
    """t^p by linear interpolation of the exponent field (t in (0, 1]); 0 for t = 0"""
    B = np.int64(0x3f800000) - np.int64(366393)          # the usual error-balancing offset
    b = t.view(np.int32).astype(np.int64)
    ub = (p * (b - B) + B).astype(np.int64)
    u = np.clip(ub, 0, 0x7f7fffff).astype(np.int32).view(np.float32)
    return np.where(t > 0, u, 0).astype(f32)


def exact_u(t, rm1):
    with np.errstate(all="ignore"):
        return np.where(t > 0, np.exp2(rm1 * np.log2(np.where(t > 0, t, 1))), 0).astype(f32)


def solve(X, alpha, approx=False, eps=0.05, theta=0.02, lin_tol=1e-4, tol=6e-7, tau_tol=2e-7, maxit=40):
    am1 = f32(alpha - 1)
    r = f32(1.0) / am1
    Xs = (X * am1).astype(f32)
    W, R, F = Xs.shape
    tau = np.maximum(Xs.max(-1) - 1, Xs.mean(-1, dtype=f32) - f32((1.0 / F) ** (alpha - 1))).astype(f32)
    n_apx = np.zeros(W, int)
    n_ex = np.zeros(W, int)
    worst_rel = 0.0
    if approx:
        active = np.ones((W, R), bool)
        for it in range(maxit):
            wa = active.any(1)
            if not wa.any():
                break
            n_apx += wa
            t = np.clip(Xs - tau[..., None], 0, None).astype(f32)
            u = pow_trick(t, float(r) - 1.0)
            ue = exact_u(t, r - 1)
            with np.errstate(all="ignore"):
                rel = np.abs(u - ue) / np.where(ue > 0, ue, 1)
            worst_rel = max(worst_rel, float(rel[t > 1e-6].max(initial=0)))
            S = (u * t).sum(-1, dtype=f32)
            D = (r * u.sum(-1, dtype=f32)).astype(f32)
            f_safe = S / f32(1 + eps) - 1
            step = f_safe / np.maximum(D / f32(1 - eps), f32(1e-30))
            act = active & (f_safe > theta) & (tau + step > tau)
            tau = np.where(act, tau + step, tau).astype(f32)
            active = act
    active = np.ones((W, R), bool)
    for it in range(maxit):
        wa = active.any(1)
        if not wa.any():
            break
        n_ex += wa
        t = np.clip(Xs - tau[..., None], 0, None).astype(f32)
        u = exact_u(t, r - 1)
        S = (u * t).sum(-1, dtype=f32)
        D = (r * u.sum(-1, dtype=f32)).astype(f32)
        f = S - 1
        step = f / np.maximum(D, f32(1e-30))
        thr = np.maximum(tol, tau_tol * min(1.0, float(am1) / 0.7) * D)
        act = active & (f > thr) & (tau + step > tau)
        lin = act & (step < lin_tol)                       # first-order finish: the step is taken, no confirming evaluation
        tau = np.where(act, tau + step, tau).astype(f32)
        active = act & ~lin
    # float64 residual at the final thresholds (left of the root <=> residual >= 0)
    t = np.clip(Xs.astype(np.float64) - tau[..., None].astype(np.float64), 0, None)
    res = (t ** float(r)).sum(-1) - 1
    return dict(apx=n_apx.mean(), ex=n_ex.mean(), tau=tau, res_max=float(res.max()), res_min=float(res.min()), rel=worst_rel)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--alphas", type=float, nargs="*", default=[1.7, 1.3, 1.9])
    ap.add_argument("--batch", type=int, default=2048)
    a = ap.parse_args()
    PAIRS, C_APX, C_EX, C_FIX = 5, 28, 72, 100            # issue cycles: per pair (approximate / exact), per evaluation (reduce, step, ballots)
    for regime in ("stress", "fresh"):
        X = gates_for(regime, a.batch)
        for alpha in a.alphas:
            base = solve(X, alpha)
            cb = base["ex"] * (PAIRS * C_EX + C_FIX)
            print(f"{regime:6s} alpha {alpha}: product solver {base['ex']:.2f} exact evaluations per pass, {cb:.0f} issue cycles "
                  f"(residual at the end in [{base['res_min']:.1e}, {base['res_max']:.1e}])")
            for eps in (0.04, 0.06):
                for theta in (0.0, 0.02, 0.1):
                    v = solve(X, alpha, approx=True, eps=eps, theta=theta)
                    c = v["apx"] * (PAIRS * C_APX + C_FIX) + v["ex"] * (PAIRS * C_EX + C_FIX)
                    tr = v["ex"] / base["ex"]
                    print(f"    approx early eps {eps} theta {theta}: {v['apx']:.2f} approximate + {v['ex']:.2f} exact evaluations, "
                          f"transcendental issues x{tr:.2f}, issue cycles {c:.0f} (x{c / cb:.2f}), trick's worst error {v['rel']:.3f}, "
                          f"max |dtau| {np.abs(v['tau'] - base['tau']).max():.1e}, residual in [{v['res_min']:.1e}, {v['res_max']:.1e}]")


if __name__ == "__main__":
    main()
