import numpy as np, sys
f=np.float32
rng=np.random.default_rng(5)
def dot(a,b): return (a[...,0]*b[...,0]+a[...,1]*b[...,1])+a[...,2]*b[...,2]
def cross(a,b): return np.stack([a[...,1]*b[...,2]-b[...,1]*a[...,2], a[...,2]*b[...,0]-b[...,2]*a[...,0], a[...,0]*b[...,1]-b[...,0]*a[...,1]],-1)
def unit(v): return v/np.linalg.norm(v,axis=-1,keepdims=True)
N=4_000_000
for sigma in (1.0,0.3):
  for lgc in np.arange(-8,-1.9,1.0):
    a=unit(rng.normal(size=(N,3))); b=unit(np.cross(a,rng.normal(size=(N,3)))); n=np.cross(a,b)
    L1=np.exp(rng.uniform(np.log(0.01),np.log(1.0),(N,1))); L2=L1*np.exp(rng.uniform(np.log(0.5),np.log(2),(N,1)))
    p1=rng.uniform(-1,1,(N,3))
    e1d=a*L1; e2d=(a*np.sqrt(1-sigma**2)+b*sigma)*L2
    p1f=p1.astype(f); p2f=(p1+e1d).astype(f); p3f=(p1+e2d).astype(f)
    e1=p2f-p1f; e2=p3f-p1f
    al=rng.uniform(-0.5,1.5,(N,1)); be=rng.uniform(-0.5,1.5,(N,1))
    tgt=p1+al*e1d+be*e2d
    # origin: in-plane displacement of length R plus normal offset R*cosphi
    R=np.exp(rng.uniform(np.log(0.05),np.log(3),(N,1))); ang=rng.uniform(0,2*np.pi,(N,1))
    cphi=10.0**(lgc+rng.uniform(0,1,(N,1)))
    o=(tgt+R*(np.cos(ang)*a+np.sin(ang)*b)+R*cphi*n*rng.choice([-1,1],(N,1))).astype(f)
    d=unit(tgt-o.astype(np.float64)).astype(f)
    q=cross(d,e2); det=dot(e1,q)
    with np.errstate(all='ignore'):
        inv=f(1.0)/det; s=o-p1f; u=inv*dot(s,q); r=cross(s,e1); v=inv*dot(d,r); t=inv*dot(e2,r)
        has=(t>f(1e-4))&~((u<0)|(u>1))&~((v<0)|(u+v>1))
    idx=np.nonzero(has)[0]
    lo=np.minimum(np.minimum(p1f[idx],p2f[idx]),p3f[idx]).astype(np.float64); hi=np.maximum(np.maximum(p1f[idx],p2f[idx]),p3f[idx]).astype(np.float64)
    O=o[idx].astype(np.float64); D=d[idx].astype(np.float64); T=t[idx].astype(np.float64)
    delta=4e-6*(np.abs(O).sum(1)+1.0)
    def viol(g):
        g=g[:,None]
        with np.errstate(all='ignore'):
            A=(lo-g-O)/D; B=(hi+g-O)/D
        t0=np.nanmax(np.minimum(A,B),axis=1); t1=np.nanmin(np.maximum(A,B),axis=1)
        t0=np.maximum(t0,0)
        return ~((t0<=t1)&(t0<=T*1.001+g[:,0])), t0, t1
    emax=np.maximum(np.maximum(np.linalg.norm(e1[idx].astype(np.float64),axis=1),np.linalg.norm(e2[idx].astype(np.float64),axis=1)),np.linalg.norm((e2[idx]-e1[idx]).astype(np.float64),axis=1))
    v2,t0,t1=viol(delta/2)
    vg,_,_=viol(delta/2+5e-4*emax/sigma)
    # true t in float64
    e1D=e1[idx].astype(np.float64); e2D=e2[idx].astype(np.float64); S=O-p1f[idx].astype(np.float64)
    Q=np.cross(D,e2D); DET=(e1D*Q).sum(1); TT=(e2D*np.cross(S,e1D)).sum(1)/DET
    rel=np.abs(T-TT)/np.abs(TT)
    print("sigma %.1f cosphi 1e%d..1e%d: hits %7d violations %5d  with the box grown by 5e-4 emax / sigma: %5d   max rel err of t %.3g  median %.3g"%(sigma,lgc,lgc+1,len(idx),v2.sum(),vg.sum(),rel.max(),np.median(rel)))
    if v2.any():
        k=np.nonzero(v2)[0][0]; print("     e.g. t_f %.6g true t %.6g t0 %.6g t1 %.6g delta %.3g"%(T[k],TT[k],t0[k],t1[k],delta[k]))
