#!/usr/bin/env python3
"""Developer tool (CPU, numpy): wave-level simulation of the sparse-map solvers on the gates of bench.py's two weight regimes.
Counts evaluations per 16-neuron pass of a 2-sample wave group (a wave runs to the slowest of its 32 rows) for the start
thresholds and step rules weighed in DESIGN.md section 9: the current start, top-k starts, a lane-local top-2 start,
sqrt- / power-transformed Newton, quadratic extrapolation, the tail-model jump, per-sample exit; alpha 2, 1.7, 1.5.
    python tools/solver_sim.py        (about two minutes)"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'arm-net_amd')):
    sys.path.insert(0, _p)
import bench
a = argparse.Namespace(gpus=1, steps=1, warmup=0, alpha=2.0, regime="stress", batch=4096, nfield=39, nfeat=1_000_000, nemb=16,
             nhid=32, nhead=1, ids="uniform", shard="replicate", micro_batches=1, rotate=1, ensemble=False)
def gates_for(regime, nhead=1):
    a.regime=regime; a.nhead=nhead
    m = bench.build_model(a, torch.device("cpu"), regime=regime)
    ids, vals, _, _ = bench.make_batch(a, 0, torch.device("cpu"), 0)
    with torch.no_grad():
        v = vals.clamp(1e-3, 1.0)
        x = m.embedding.embedding.weight[ids] * v.unsqueeze(2)          # B,F,E
        if nhead==1:
            W = m.attn_layer.bilinear_w.weight; q = m.attn_layer.query
            qf = (q @ W) * (a.nemb ** -0.5)                                  # O,E
        else:
            W = m.attn_layer.bilinear_w; q = m.attn_layer.query  # K,E,D ; K,H,D
            qf = torch.einsum('kxy,koy->kox', W, q).reshape(-1, a.nemb) * (a.nemb ** -0.5)
        g = torch.einsum('bfe,oe->bof', x, qf)
    return g.numpy().astype(np.float32)

def michelot_wave(g, tau0_fn, tol=6e-7, maxit=40, per_sample_skip=False):
    # g: [B,O,F]; waves = (2 samples, 16 neurons); returns avg evaluations per wave-pass, and avg sample-evals
    B,O,F = g.shape
    X = g.reshape(B//2, 2, O//16, 16, F).transpose(0,2,1,3,4).reshape(-1, 32, F)   # [waves, 32 rows, F]
    tau = tau0_fn(X)
    active = np.ones(X.shape[:2], bool)
    evals = np.zeros(X.shape[0], int)
    sevals = np.zeros(X.shape[0], float)
    for it in range(maxit):
        wave_act = active.any(1)
        if not wave_act.any(): break
        evals += wave_act
        # per-sample: rows 0..15 sample0, 16..31 sample1
        sa = active.reshape(-1,2,16).any(2)
        sevals += sa.sum(1) if per_sample_skip and it>0 else 2*wave_act
        t = np.clip(X - tau[...,None], 0, None)
        S = t.sum(-1); k = (t>0).sum(-1)
        f = S - 1
        tn = tau + f / np.maximum(k,1)
        act = (f > tol) & (tn > tau)
        act &= active   # rows go passive permanently? in kernel: act recomputed each eval; passive rows have f<=tol always
        tau = np.where(act, tn, tau)
        active = act
    return evals.mean(), sevals.mean()/2

def tau_cur(X):
    mx = X.max(-1); mean = X.mean(-1)
    F = X.shape[-1]
    return np.maximum(mx - 1, mean - 1.0/F)
def tau_top2(X):
    s = -np.sort(-X, axis=-1)
    c = np.cumsum(s[..., :2], -1)
    t2 = (c[...,1]-1)/2
    return np.maximum(tau_cur(X), t2)
def tau_topk(k):
    def fn(X):
        s = -np.sort(-X, axis=-1)
        c = np.cumsum(s[..., :k], -1)
        cand = (c - 1)/np.arange(1,k+1)
        return np.maximum(tau_cur(X), cand.max(-1))
    return fn
def tau_lane_top2(X):
    # lane g holds fields 4j+g: lane-local top2 of 10 -> 8 candidates; tau = max over k of prefix of sorted 8 candidates
    W,R,F = X.shape
    Xp = np.concatenate([X, np.full((W,R,40-F), -np.inf, X.dtype)], -1).reshape(W,R,10,4)  # [.., j, g]
    s = -np.sort(-Xp, axis=2)[:, :, :2, :]     # top2 per lane
    c8 = -np.sort(-s.reshape(W,R,8), axis=-1)
    c = np.cumsum(c8, -1)
    cand = (c-1)/np.arange(1,9)
    return np.maximum(tau_cur(X), cand.max(-1))
def tau_exact(X):
    s = -np.sort(-X, axis=-1); c=np.cumsum(s,-1); k=np.arange(1,X.shape[-1]+1)
    return ((c-1)/k).max(-1)

for regime in ("fresh","stress"):
    g = gates_for(regime)
    sup = (np.clip(g - tau_exact(g.reshape(-1,1,g.shape[-1]))[...,None].reshape(g.shape[0],g.shape[1],1),0,None)>0).sum(-1)
    print(regime, "support mean", sup.mean(), "p50", np.median(sup), "p99", np.percentile(sup,99), "gate std", g.std())
    for name, fn in (("current", tau_cur), ("top2", tau_top2), ("top3", tau_topk(3)), ("top4",tau_topk(4)), ("lane_top2(8 cand)", tau_lane_top2), ("exact", tau_exact)):
        e, se = michelot_wave(g, fn)
        e2, se2 = michelot_wave(g, fn, per_sample_skip=True)
        print(f"  {name:20s} evals/wave-pass {e:.2f}   with per-sample skip: sample-evals {se2:.2f}")

print("---- sqrt-Newton variants (alpha=2)")
def solve_wave(g, tau0_fn, step_fn, tol=6e-7, maxit=40):
    B,O,F = g.shape
    X = g.reshape(B//2, 2, O//16, 16, F).transpose(0,2,1,3,4).reshape(-1, 32, F)
    tau = tau0_fn(X)
    active = np.ones(X.shape[:2], bool)
    evals = np.zeros(X.shape[0], int)
    rows_evals = np.zeros(X.shape[:2], int)
    for it in range(maxit):
        wave_act = active.any(1)
        if not wave_act.any(): break
        evals += wave_act
        rows_evals += active
        t = np.clip(X - tau[...,None], 0, None)
        S = t.sum(-1); k = np.maximum((t>0).sum(-1),1)
        tn = step_fn(tau, S, k, it)
        mx = X.max(-1)
        tn = np.minimum(tn, mx - 1e-6)   # keep k>=1
        tn = np.maximum(tn, mx-1)
        act = (np.abs(S-1) > tol) & (tn != tau) & active
        tau = np.where(act, tn, tau)
        active = act
    # check result
    ex = tau_exact(X)
    err = np.abs(tau-ex).max()
    return evals.mean(), rows_evals.mean(), err
def newton(tau,S,k,it): return tau + (S-1)/k
def sqrtn(thr):
    def fn(tau,S,k,it):
        rs = np.sqrt(np.maximum(S,0))
        st = np.where(S>thr, 2*rs*(rs-1)/k, (S-1)/k)
        return tau+st
    return fn
def sqrtn_first(n):
    def fn(tau,S,k,it):
        rs = np.sqrt(np.maximum(S,0))
        st = np.where((S>1)&(it<n), 2*rs*(rs-1)/k, (S-1)/k)
        return tau+st
    return fn
g = gates_for("stress")
for name, t0, st in (("newton/cur", tau_cur, newton), ("sqrt thr=1", tau_cur, sqrtn(1.0)), ("sqrt thr=1.05", tau_cur, sqrtn(1.05)),("sqrt thr=1.2", tau_cur, sqrtn(1.2)), ("sqrt thr=1.5", tau_cur, sqrtn(1.5)), ("sqrt thr=2", tau_cur, sqrtn(2.0)),
     ("sqrt first1", tau_cur, sqrtn_first(1)), ("sqrt first2", tau_cur, sqrtn_first(2)), ("sqrt first3", tau_cur, sqrtn_first(3)),
     ("lanetop2+sqrt1.2", tau_lane_top2, sqrtn(1.2))):
    e, re, err = solve_wave(g, t0, st)
    print(f"  {name:20s} evals/wave-pass {e:.2f}  per-row {re:.2f} err {err:.2e}")

print("---- model-based steps with memory (alpha=2)")
def solve_wave2(X, tau0_fn, strat, tol=6e-7, maxit=40):
    tau = tau0_fn(X)
    mx = X.max(-1)
    active = np.ones(X.shape[:2], bool)
    evals = np.zeros(X.shape[0], int)
    rows_evals = np.zeros(X.shape[:2], int)
    pt = np.full(tau.shape, np.nan); pS = np.full(tau.shape, np.nan); pk = np.full(tau.shape, np.nan)
    for it in range(maxit):
        wave_act = active.any(1)
        if not wave_act.any(): break
        evals += wave_act
        rows_evals += active
        t = np.clip(X - tau[...,None], 0, None)
        S = t.sum(-1); k = np.maximum((t>0).sum(-1),1).astype(np.float64)
        tn = strat(tau, S, k, pt, pS, pk, it)
        tn = np.minimum(tn, mx - 1e-6)
        tn = np.maximum(tn, mx-1)
        act = (np.abs(S-1) > tol) & (tn != tau) & active
        pt = np.where(act, tau, pt); pS = np.where(act, S, pS); pk = np.where(act, k, pk)
        tau = np.where(act, tn, tau)
        active = act
    ex = tau_exact(X)
    return evals.mean(), rows_evals.mean(), np.abs(tau-ex).max(), np.bincount(evals)
def strat_quad(thr=1.0, first="sqrt"):
    def fn(tau,S,k,pt,pS,pk,it):
        newt = tau + (S-1)/k
        rs = np.sqrt(np.maximum(S,0)); sq = tau + 2*rs*(rs-1)/k
        with np.errstate(all='ignore'):
            rho = (pk - k)/(tau - pt)                 # elements per unit tau between the last two points (>=0 when moving right)
            disc = k*k - 2*rho*(S-1)
            d = (k - np.sqrt(np.maximum(disc,0)))/rho
            quad = tau + d
        ok = np.isfinite(quad) & (rho > 1e-6) & (disc > 0) & (S > thr) & (tau > pt)
        out = np.where(ok, quad, newt)
        if first == "sqrt":
            out = np.where((it==0) & (S>1), sq, out)
        return out
    return fn
Xs = g.reshape(g.shape[0]//2, 2, g.shape[1]//16, 16, g.shape[2]).transpose(0,2,1,3,4).reshape(-1, 32, g.shape[2])
for name, st in (("newton", lambda tau,S,k,pt,pS,pk,it: tau+(S-1)/k), ("quad thr1.0", strat_quad(1.0)), ("quad thr1.02", strat_quad(1.02)), ("quad thr1.1", strat_quad(1.1)), ("quad thr1.3", strat_quad(1.3)), ("quad no-sqrt-first", strat_quad(1.02, first=None))):
    e, re, err, hist = solve_wave2(Xs, tau_cur, st)
    print(f"  {name:20s} evals/wave-pass {e:.2f}  per-row {re:.2f} err {err:.2e} hist {hist}")

print("---- beta-tail model jump (alpha=2)")
def strat_beta(thr_lo=1.1, nmax=9, damp=1.0):
    def fn(tau,S,k,pt,pS,pk,it):
        newt = tau + (S-1)/k
        M = mxg - tau
        with np.errstate(all='ignore'):
            m1 = S/k
            beta = np.clip(M/m1 - 1, 0.2, 8.0)
            rem = np.exp2((np.log2((beta+1)/k) + beta*np.log2(M))/(beta+1))     # M - d
            jump = tau + damp*(M - rem)
        ok = np.isfinite(jump) & (S > thr_lo) & (it < nmax) & (jump > newt)
        return np.where(ok, jump, newt)
    return fn
mxg = Xs.max(-1)
for name, st in (("beta thr1.1", strat_beta(1.1)), ("beta thr1.02", strat_beta(1.02)), ("beta thr1.3", strat_beta(1.3)), ("beta thr1.1 first only", strat_beta(1.1, nmax=1)), ("beta thr1.1 first2", strat_beta(1.1, nmax=2)), ("beta damp.9", strat_beta(1.1, damp=0.9)), ("beta thr 2", strat_beta(2.0)), ("beta thr 1.6", strat_beta(1.6))):
    e, re, err, hist = solve_wave2(Xs, tau_cur, st)
    print(f"  {name:24s} evals/wave-pass {e:.2f}  per-row {re:.2f} err {err:.2e} hist {hist}")

print("---- generic alpha Newton (alpha=1.7)")
def newton_alpha(X, alpha, tol=6e-7, maxit=40, transform=None, per_sample=False):
    am1 = np.float32(alpha-1); r = 1.0/(alpha-1)
    Xs_ = X*am1
    F = X.shape[-1]
    mx = Xs_.max(-1); mean = Xs_.mean(-1)
    tau = np.maximum(mx-1, mean - (1.0/F)**(alpha-1))
    active = np.ones(X.shape[:2], bool)
    evals = np.zeros(X.shape[0], int); sev = np.zeros(X.shape[0], float)
    for it in range(maxit):
        wa = active.any(1)
        if not wa.any(): break
        evals += wa
        sev += active.reshape(-1,2,16).any(2).sum(1)/2
        t = np.clip(Xs_-tau[...,None],0,None)
        u = np.where(t>0, t**(r-1), 0); S = (u*t).sum(-1); D = r*u.sum(-1)
        f = S-1
        if transform == "pow" and True:
            # Newton on S^(1/(r+1)) - 1
            e = 1.0/(r+1)
            h = S**e - 1; hp = -e*S**(e-1)*D
            step = np.where(S>1.0, -h/hp, f/D)
        else:
            step = f/np.maximum(D,1e-30)
        tn = tau+step
        act = (f>tol)&(tn>tau)&active
        tau = np.where(act,tn,tau); active = act
    return evals.mean(), sev.mean()
for alpha in (1.7, 1.5):
    for name, tr in (("plain",None),("pow-transform","pow")):
        e, se = newton_alpha(Xs, alpha, transform=tr)
        print(f"  alpha {alpha} {name:15s} evals/wave-pass {e:.2f}  per-sample-exit {se:.2f}")

print("---- generic alpha: two-sided Newton with model jumps")
def newton2(X, alpha, mode, thr=1.2, tol=6e-7, maxit=40):
    am1 = np.float32(alpha-1); r = 1.0/(alpha-1)
    Xs_ = (X*am1).astype(np.float64)
    F = X.shape[-1]
    mx = Xs_.max(-1); mean = Xs_.mean(-1)
    tau = np.maximum(mx-1, mean - (1.0/F)**(alpha-1))
    active = np.ones(X.shape[:2], bool)
    evals = np.zeros(X.shape[0], int)
    for it in range(maxit):
        wa = active.any(1)
        if not wa.any(): break
        evals += wa
        t = np.clip(Xs_-tau[...,None],0,None)
        if alpha == 2.0:
            S = t.sum(-1); D = (t>0).sum(-1).astype(np.float64)
        else:
            u = np.where(t>0, t**(r-1), 0); S = (u*t).sum(-1); D = r*u.sum(-1)
        f = S-1
        newt = tau + f/np.maximum(D,1e-300)
        with np.errstate(all='ignore'):
            if mode == "pow":
                e = 1.0/(r+1)
                jump = tau + (S - S**(1-e))/(e*D)
            elif mode == "tail":
                M = mx - tau
                n = D*M/S
                jump = mx - M*S**(-1.0/n)
            else:
                jump = newt
        use = (S > thr) & np.isfinite(jump) & (jump > newt)
        tn = np.where(use, jump, newt)
        tn = np.where(tn >= mx, 0.5*(tau+mx), tn)
        tn = np.maximum(tn, mx-1)
        act = (np.abs(f)>tol)&(tn!=tau)&active
        tau = np.where(act,tn,tau); active = act
    # exactness check vs bisection-free reference: f at final tau
    t = np.clip(Xs_-tau[...,None],0,None)
    S = t.sum(-1) if alpha==2.0 else (np.where(t>0,t**r,0)).sum(-1)
    return evals.mean(), np.abs(S-1).max(), np.bincount(evals)
for alpha in (2.0, 1.7, 1.5):
    for mode in ("plain","pow","tail"):
        for thr in ((1.2,) if mode=="plain" else (1.05, 1.2, 1.5)):
            e, res, hist = newton2(Xs, alpha, mode, thr)
            print(f"  alpha {alpha} {mode:6s} thr {thr}: evals/wave-pass {e:.2f}  max|S-1| {res:.1e} hist {hist[3:]}")
