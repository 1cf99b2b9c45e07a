import numpy as np, sys
f=np.float32
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
def dot(a,b): return (a[...,0]*b[...,0]+a[...,1]*b[...,1])+a[...,2]*b[...,2]
def cross(a,b): return np.stack([a[...,1]*b[...,2]-b[...,1]*a[...,2], a[...,2]*b[...,0]-b[...,2]*a[...,0], a[...,0]*b[...,1]-b[...,0]*a[...,1]],-1)
def unit(v): return v/np.linalg.norm(v,axis=-1,keepdims=True)
N=2_000_000
res={}
for lg in np.arange(-8,0.01,0.5):
    sigma=10.0**lg
    a=unit(rng.normal(size=(N,3))); b=unit(np.cross(a,rng.normal(size=(N,3))))
    L1=np.exp(rng.uniform(np.log(0.01),np.log(1.0),(N,1))); L2=L1*np.exp(rng.uniform(np.log(0.3),np.log(3),(N,1)))
    p1=rng.uniform(-1,1,(N,3))
    e1d=a*L1; e2d=(a*np.sqrt(1-sigma**2)+b*sigma)*L2
    p1f=p1.astype(f); p2f=(p1+e1d).astype(f); p3f=(p1+e2d).astype(f)
    e1=p2f-p1f; e2=p3f-p1f
    o=rng.uniform(-2,2,(N,3)).astype(f)
    al=rng.uniform(-3,4,(N,1)); be=rng.uniform(-3,4,(N,1))
    # half of the targets along the supporting line / plane, tiny offsets
    off=unit(rng.normal(size=(N,3)))*np.exp(rng.uniform(np.log(1e-9),np.log(1e-2),(N,1)))
    tgt=p1+al*e1d+be*e2d*(rng.random((N,1))<0.5)+off
    d=unit(tgt-o.astype(np.float64)).astype(f)
    q=cross(d,e2); det=dot(e1,q)
    with np.errstate(all='ignore'):
        inv=f(1.0)/det; s=o-p1f; u=inv*dot(s,q); r=cross(s,e1); v=inv*dot(d,r); t=inv*dot(e2,r)
        has=(t>f(1e-4))&~((u<0)|(u>1))&~((v<0)|(u+v>1))
    idx=np.nonzero(has)[0]
    if len(idx)==0: print(lg,'no hits'); continue
    P=o[idx].astype(np.float64)+d[idx].astype(np.float64)*t[idx].astype(np.float64)[:,None]
    lo=np.minimum(np.minimum(p1f[idx],p2f[idx]),p3f[idx]).astype(np.float64); hi=np.maximum(np.maximum(p1f[idx],p2f[idx]),p3f[idx]).astype(np.float64)
    exc=np.maximum(np.maximum(lo-P,P-hi),0).max(1)
    delta=4e-6*(np.abs(o[idx].astype(np.float64)).sum(1)+1.0)
    emax=np.maximum(np.linalg.norm(e1[idx],axis=1),np.linalg.norm(e2[idx],axis=1))
    # also: sigma_eff from float edges
    cr=np.linalg.norm(np.cross(e1[idx].astype(np.float64),e2[idx].astype(np.float64)),axis=1); sg=cr/(np.linalg.norm(e1[idx].astype(np.float64),axis=1)*np.linalg.norm(e2[idx].astype(np.float64),axis=1))
    O=o[idx].astype(np.float64); D=d[idx].astype(np.float64); T=t[idx].astype(np.float64)
    def viol(g):
        g=g[:,None]
        with np.errstate(all='ignore'):
            a=(lo-g-O)/D; b=(hi+g-O)/D
        t0=np.nanmax(np.minimum(a,b),axis=1); t1=np.nanmin(np.maximum(a,b),axis=1)
        t0=np.maximum(t0,0)
        return ~((t0<=t1)&(t0<=T*1.001+g[:,0]))
    v2=viol(delta/2); v8=viol(delta/8)
    nrm=np.cross(e1[idx].astype(np.float64),e2[idx].astype(np.float64)); cosphi=np.abs((unit(nrm)*D).sum(1))
    print("   violations at delta/2: %d  at delta/8: %d   (of %d hits); violators' cosphi min/median %s"%(v2.sum(),v8.sum(),len(idx), (cosphi[v2].min(),np.median(cosphi[v2])) if v2.any() else '-'))
    ratio=exc/delta
    k=np.argmax(ratio)
    print("sigma 1e%.1f hits %7d  max exc/delta %10.3g  (exc %.3g, emax %.3g, sg %.3g, t %.3g) frac>0.25: %.2e  max exc*sg/emax/delta %.3g"%(lg,len(idx),ratio.max(),exc[k],emax[k],sg[k],t[idx][k],(ratio>0.25).mean(), (ratio*sg/emax).max()))
