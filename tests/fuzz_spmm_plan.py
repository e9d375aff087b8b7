"""One-off fuzz of the SpMM schedule builders (gdmcf_amd/lightgcn.py: spmm_bundle_plan + spmm_stream_pack) on the CPU: random CSR
matrices with degenerate degree patterns (all rows empty / full, hubs, tiny), random widths and limits; the packed stream is
walked in numpy the way the kernel walks it and must reproduce A @ X with every row written exactly once.
    python tests/fuzz_spmm_plan.py"""
import numpy as np, scipy.sparse as sp, sys, traceback
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gdmcf_amd.lightgcn import spmm_bundle_plan, spmm_stream_pack

def emulate(A, X, d, sp_):
    n=A.shape[0]; G,UN,DW=sp_["G"],sp_["UN"],sp_["DW"]
    cw=sp_["cw"].reshape(-1,2); cc=cw[:,0]; ww=cw[:,1].copy().view(np.float32)
    ud=sp_["ud"].reshape(-1,DW); wd=sp_["wdesc"].reshape(-1,4)
    Y=np.full((n,d),np.nan); partial=np.full((max(sp_["n_slots"],1),d),np.nan); written=np.zeros(n,int)
    n_batches=len(cw)//64
    for w in range(len(wd)):
        sb,nb,u0,u1=wd[w]; pos=sb*64
        # the kernel pre-loads batch 0 and batch min(1, nb-1) of the run of every wave that has units: they must exist
        assert u1<=u0 or (nb>=1 and sb+nb<=n_batches), ("wave with units but no batch", w, sb, nb, u0, u1)
        for u in range(u0,u1):
            hdr=int(ud[u,0]); ng=hdr&0x7FFFFFFF
            acc=np.zeros((G,d))
            for _ in range(ng*UN):
                for g in range(G):
                    acc[g]+=float(ww[pos+g])*X[cc[pos+g]]
                pos+=G
            if hdr<0:
                row,slot=ud[u,1],ud[u,2]
                if slot>=0: partial[slot]=acc.sum(0)
                else: Y[row]=acc.sum(0); written[row]+=1
            else:
                for g in range(G):
                    a=int(ud[u,1+g])
                    if a<0: continue
                    r=a&0x3FFFFFFF; Y[r]=0.0 if a&0x40000000 else acc[g]; written[r]+=1
    for i,r in enumerate(sp_["crow"]):
        Y[r]=partial[sp_["cptr"][i]:sp_["cptr"][i+1]].sum(0); written[r]+=1
    assert (written==1).all(), written
    return Y

rng=np.random.default_rng(0)
cases=0
for trial in range(400):
    n=int(rng.integers(1,60)); m=int(rng.integers(1,80)); d=int(rng.choice([8,16,64,256]))
    kind=rng.integers(0,6)
    if kind==0: deg=np.zeros(n,int)
    elif kind==1: deg=np.full(n,m)
    elif kind==2: deg=rng.integers(0,min(m,3)+1,n)
    elif kind==3: deg=np.minimum(rng.zipf(1.3,n),m)
    elif kind==4: deg=np.where(rng.random(n)<0.5,0,m)
    else: deg=rng.integers(0,m+1,n)
    rows=np.repeat(np.arange(n),deg)
    cols=np.concatenate([np.sort(rng.choice(m,k,replace=False)) for k in deg]) if deg.sum() else np.zeros(0,int)
    A=sp.csr_matrix((rng.standard_normal(len(rows)).astype(np.float32),(rows,cols)),shape=(n,m)); A.sort_indices()
    if A.nnz==0: continue
    X=rng.standard_normal((m,d))
    try:
        for s_max,piece,nw in ((int(rng.integers(0,20)),int(rng.integers(1,40)),[32,64,None][int(rng.integers(0,3))]),):
            pl=spmm_bundle_plan(A.indptr,A.indices,d=d,n_waves=nw,s_max=s_max,piece=piece,n_cols=m)
            sp_=spmm_stream_pack(pl,A.indptr,A.indices,A.data,d=d)
            Y=emulate(A,X,d,sp_)
            np.testing.assert_allclose(Y,A@X,rtol=1e-5,atol=1e-5)
            cases+=1
    except Exception as e:
        print("FAIL trial",trial,"n",n,"m",m,"d",d,"kind",kind,"smax",s_max,"piece",piece,"nw",nw); traceback.print_exc(); sys.exit(1)
print("ok",cases)
