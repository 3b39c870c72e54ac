import numpy as np
L=np.arange(64); G=L>>4; C=L&15
def mfma(a,b,c):
    # a,b: [64] ; c: [64,4]; D[i][j] += sum_slot a(i,slot) b(slot,j); lane(g,c): a[i=c][slot=g], b[slot=g][j=c]; D row=g+4r, col=c
    Am=np.zeros((16,4)); Bm=np.zeros((4,16))
    for l in range(64): Am[C[l],G[l]]=a[l]; Bm[G[l],C[l]]=b[l]
    P=Am@Bm
    out=c.copy()
    for l in range(64):
        for r in range(4): out[l,r]+=P[G[l]+4*r,C[l]]
    return out
def to_regs(T):   # 16x16 -> [64,4]
    out=np.zeros((64,4))
    for l in range(64):
        for r in range(4): out[l,r]=T[G[l]+4*r,C[l]]
    return out
def from_regs(R):
    T=np.zeros((16,16))
    for l in range(64):
        for r in range(4): T[G[l]+4*r,C[l]]=R[l,r]
    return T
def prod(acc,X,Y):
    for r in range(4): acc=mfma(X[:,r],Y[:,r],acc)
    return acc
def base(A,klim):
    X=np.zeros((64,4))
    for r in range(4): X[:,r]=(G+4*r==C)*1.0
    XT=X.copy(); pv=np.ones(64); ipv=np.ones(64)
    p=A[0,0]; ip=1/p
    for kr in range(4):
        for kg in range(4):
            k=4*kr+kg
            if k<klim:
                uk=A[:,kr].copy(); xk=X[:,kr].copy()
                pv=np.where(C==k,p,pv); ipv=np.where(C==k,ip,ipv)
                pn=ipn=1.0
                if k+1<klim:
                    b=uk[16*kg+k+1]
                    un=A[:,kr+1] if kg==3 else uk
                    a=un[16*((kg+1)&3)+k+1]
                    pn=a-b*ip*b; ipn=1/pn
                rowk=G==kg; act=rowk&(C>k)
                aop=np.where(act,-ip*uk,0.0); bopA=np.where(act,uk,0.0); bopX=np.where(rowk,xk,0.0)
                A=mfma(aop,bopA,A); X=mfma(aop,bopX,X); XT=mfma(bopX,aop,XT)
                p,ip=pn,ipn
    return X,XT,pv,ipv

def run(Afull,npiv):
    n=64
    nbase=(npiv+15)>>4
    Ain=Afull.copy()
    for R in range(64):
        for Cc in range(64):
            if R>=npiv or Cc>=npiv: Ain[R,Cc]=1.0 if R==Cc else 0.0
    Areg={}  # slot (I,J) -> 16x16
    for I in range(4):
        for J in range(4): Areg[(I,J)]=Ain[16*I:16*I+16,16*J:16*J+16].copy()
    Xreg={}; MT={}; pl=np.ones(64); ipl=np.ones(64)
    flags=set()
    def ld(d,k): return to_regs(d[k])
    def ld_scaled(slot,s):
        h=to_regs(Areg[slot]); out=h.copy()
        for r in range(4): out[:,r]=-h[:,r]*ipl[16*s+G+4*r]
        return out
    def wave0():
        X=XT=MTs=None
        for p in range(nbase):
            if p==0: A=ld(Areg,(0,0))
            else:
                while ('SUP',p) not in flags: yield
                S=ld(Areg,(p-1,p)); Dg=ld(Areg,(p,p))
                H=prod(np.zeros((64,4)),XT,S); Hs=prod(np.zeros((64,4)),MTs,S)
                Areg[(p,p-1)]=from_regs(H); flags.add(('HF',p-1,p))
                A=prod(Dg,Hs,H)
            klim=min(16,npiv-16*p)
            X,XT,pv,ipv=base(A,klim)
            MTs=-XT*ipv[:,None]
            Xreg[(p,p)]=from_regs(X); MT[p]=from_regs(XT)
            for l in range(64):
                if G[l]==0: pl[16*p+C[l]]=pv[l]; ipl[16*p+C[l]]=ipv[l]
            flags.add(('BASE',p))
            yield
    def wavec(cc):
        if not (cc<nbase): return
        for q in range(cc):
            acc=ld(Areg,(q,cc))
            for s in range(q):
                while ('HF',s,q) not in flags: yield
                acc=prod(acc,ld_scaled((q,s),s),ld(Areg,(cc,s)))
            if q==cc-1:
                Areg[(q,cc)]=from_regs(acc)
                dg=ld(Areg,(cc,cc))
                for s in range(q): dg=prod(dg,ld_scaled((cc,s),s),ld(Areg,(cc,s)))
                Areg[(cc,cc)]=from_regs(dg); flags.add(('SUP',cc))
            else:
                while ('BASE',q) not in flags: yield
                Areg[(cc,q)]=from_regs(prod(np.zeros((64,4)),to_regs(MT[q]),acc)); flags.add(('HF',q,cc))
        for jj in range(cc):
            if jj==cc-1:
                while ('HF',jj,cc) not in flags: yield
            if jj>0:
                while ('YROW',jj) not in flags: yield
            hs=ld_scaled((cc,jj),jj)
            for p in range(jj+1):
                t0=np.zeros((64,4)) if jj==p else ld(Xreg,(cc,p))
                Xreg[(cc,p)]=from_regs(prod(t0,hs,ld(Xreg,(jj,p))))
        while ('BASE',cc) not in flags: yield
        mt=to_regs(MT[cc])
        for p in range(cc):
            Xreg[(cc,p)]=from_regs(prod(np.zeros((64,4)),mt,ld(Xreg,(cc,p))))
        flags.add(('YROW',cc))
    gens=[wave0(),wavec(1),wavec(2),wavec(3)]
    alive=[True]*4
    for it in range(10000):
        if not any(alive): break
        for i,gn in enumerate(gens):
            if alive[i]:
                try: next(gn)
                except StopIteration: alive[i]=False
    assert not any(alive)
    Xo=np.zeros((64,64))
    for R in range(64):
        I=R>>4; rs=1/np.sqrt(pl[R])
        for Cc in range(64):
            J=Cc>>4
            if R>=npiv: Xo[R,Cc]=1.0 if R==Cc else 0.0
            elif Cc<=R: Xo[R,Cc]=Xreg[(I,J)][R&15,Cc&15]*rs
    return Xo,pl
rng=np.random.default_rng(1)
B=rng.uniform(-.5,.5,(64,64)); A=B@B.T+0.5*np.eye(64)
for npiv in (64,50,36,17,16,8,1):
    Xo,pl=run(A,npiv)
    Ap=np.eye(64); Ap[:npiv,:npiv]=A[:npiv,:npiv]
    Lref=np.linalg.cholesky(Ap); Xref=np.linalg.inv(Lref)
    print(npiv, np.abs(Xo-Xref).max(), np.abs(Xo@Ap@Xo.T-np.eye(64)).max())
