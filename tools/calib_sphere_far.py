import numpy as np
f=np.float32
rng=np.random.default_rng(3)
def dot(a,b): return (a[...,0]*b[...,0]+a[...,1]*b[...,1])+a[...,2]*b[...,2]
def unit(v): return v/np.linalg.norm(v,axis=-1,keepdims=True)
N=4_000_000
for lr in (-4,-3,-2,-1,0):
  for dist in (1.0,10.0,100.0):
    r=(10.0**(lr+rng.uniform(0,1,(N,1)))).astype(f)
    c=rng.uniform(-1,1,(N,3)).astype(f)
    o=(c+unit(rng.normal(size=(N,3)))*dist*rng.uniform(0.3,1,(N,1))).astype(f)
    # aim near the limb: impact parameter r*(1+x), x in +-1e-1..1e-7
    dirc=unit(c.astype(np.float64)-o.astype(np.float64)); perp=unit(np.cross(dirc,rng.normal(size=(N,3))))
    x=np.exp(rng.uniform(np.log(1e-8),np.log(1e-1),(N,1)))*rng.choice([-1,1],(N,1))
    tgt=c.astype(np.float64)+perp*r.astype(np.float64)*(1+x)
    d=unit(tgt-o.astype(np.float64)).astype(f)
    oc=o-c
    with np.errstate(all='ignore'):
        a=dot(d,d); hb=dot(oc,d); cc=dot(oc,oc)-r[:,0]*r[:,0]; disc=hb*hb-a*cc; sq=np.sqrt(disc); t0=(-hb-sq)/a; t1=(-hb+sq)/a
        t=np.where(t0>0,t0,t1); has=(disc>=0)&(t>f(1e-4))     # approximate form of rt.comp:126-146
    idx=np.nonzero(has)[0]
    O=o[idx].astype(np.float64); D=d[idx].astype(np.float64); T=t[idx].astype(np.float64); C=c[idx].astype(np.float64); R=r[idx].astype(np.float64)
    lo=C-R; hi=C+R
    delta=4e-6*(np.abs(O).sum(1)+1.0)
    def viol(g):
        g=g[:,None]
        with np.errstate(all='ignore'):
            A=(lo-g-O)/D; B=(hi+g-O)/D
        s0=np.nanmax(np.minimum(A,B),axis=1); s1=np.nanmin(np.maximum(A,B),axis=1); s0=np.maximum(s0,0)
        return ~((s0<=s1)&(s0<=T*1.001+g[:,0]))
    v=viol(delta/2)
    # true miss distance
    b=np.linalg.norm(np.cross(C-O,D),axis=1)/np.linalg.norm(D,axis=1); exc=(b-R[:,0])/R[:,0]
    print("r 1e%d..1e%d dist %5.0f: hits %7d box violations %5d  max (impact-r)/r among hits %.3g  in units of delta %.3g"%(lr,lr+1,dist,len(idx),v.sum(),exc.max(), ((b-R[:,0])/delta).max()))
