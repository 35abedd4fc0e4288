import numpy as np, sys
f32=np.float32
def run(X, alpha, n_iter=50, KMAX=8, TH=f32(4e-5), AIM=f32(1e-4), verbose=False):
    am1=f32(alpha-1); rr=f32(1)/am1; rm1=rr-f32(1)
    d=X.shape[1]
    mx=X.max(1); tau=(mx-f32(1)).astype(f32); tau_hi=(mx-f32((1.0/d)**float(am1))).astype(f32)
    def S_lit(t_at):
        t=np.clip(X-t_at[:,None],0,None).astype(f32)
        with np.errstate(divide='ignore'):
            p=np.exp2(rr*np.log2(t)).astype(f32)
        return p.sum(1,dtype=f32)
    f_lo=S_lit(tau)-f32(1); dm=(tau_hi-tau).astype(f32)
    ba=np.where(f_lo>=TH,tau,-np.inf).astype(f32); bb=np.full_like(tau,np.inf)
    fa=np.where(f_lo>=TH,f_lo,np.inf).astype(f32); fb=np.full_like(tau,-np.inf)
    L=tau.copy(); U=tau_hi.copy(); tq=(tau_hi if rr<1 else tau).copy()
    nprobe=0
    for k in range(KMAX):
        t=np.clip(X-tq[:,None],0,None).astype(f32)
        with np.errstate(divide='ignore'):
            u=np.where(t>0,np.exp2(rm1*np.log2(np.where(t>0,t,1))),0).astype(f32)
        S=(u*t).sum(1,dtype=f32); D=rr*u.sum(1,dtype=f32); f=S-f32(1)
        nprobe+=1
        up=(f>=TH)&(tq>ba); ba=np.where(up,tq,ba); fa=np.where(up,f,fa)
        dn=(f<=-TH)&(tq<bb); bb=np.where(dn,tq,bb); fb=np.where(dn,f,fb)
        L=np.where(f>0,np.maximum(L,tq),L); U=np.where(f<0,np.minimum(U,tq),U)
        done=(fa<=4*AIM)&(fb>=-4*AIM)
        if done.all(): break
        target=np.where(f<0,AIM,-AIM).astype(f32)
        tn=(tq+(f-target)/D).astype(f32)
        tn=np.where((tn>L)&(tn<U),tn,f32(0.5)*(L+U)).astype(f32)
        tq=tn
    evals=0; steps=0
    for it in range(n_iter):
        dm=(dm*f32(0.5)).astype(f32); tm=(tau+dm).astype(f32)
        up=(f_lo==0)|(tm<=ba); dn=(f_lo>0)&(tm>=bb)
        live=(tm!=tau).any()
        if (~(up|dn)).any():
            evals+=1
            f_m=S_lit(tm)-f32(1)
            ba=np.where(f_m>=TH,np.maximum(ba,tm),ba); bb=np.where(f_m<=-TH,np.minimum(bb,tm),bb)
            tau=np.where(f_m*f_lo>=0,tm,tau)
        else:
            tau=np.where(up,tm,tau)
        steps+=1
        if not live: break
    return nprobe, evals, steps
rng=np.random.default_rng(0)
for alpha in (2.5,3.0,2.2,2.0,1.7,1.5):
  for name,scale in (("dense",1e-3),("mid",0.5),("sparse",3.0),("vsparse",10.0)):
    tot=np.zeros(3)
    for w in range(50):
        g=(rng.standard_normal((16,39))*scale).astype(f32)
        tot+=run(g*f32(alpha-1),alpha)
    tot/=50
    print("alpha %.1f %-8s probes %.1f  real evals %.1f  of %.1f steps   cost %.1f vs %.1f" % (alpha,name,tot[0],tot[1],tot[2], 1+1.15*tot[0]+tot[1]+1, 1+tot[2]))
