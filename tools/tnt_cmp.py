import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from cora_amd import capi, host
from oracle import oracle as orc, tnt as otnt
for (d,n,p,loops) in [(3,150,3,0),(3,150,5,6),(2,200,3,5)]:
    P = host.Problem.synthetic(dim=d, n_poses=n, n_landmarks=3, n_ranges=n//2, n_loops=loops, seed=21)
    P.update(); P.set_rank(p)
    dm=P.dims(); _,_,rp,ci,va=P.matrix("DataMatrix")
    Q=orc.CSR(rp,ci,va,dm["N"]); dims=orc.Dims(dm["d"],dm["n"],dm["r"],dm["N"])
    x0=orc.project_manifold(dims,np.random.default_rng(2).uniform(-1,1,(dims.N,p)))
    got=P.tnt(x0,max_seconds=120); ref=otnt.tnt(Q,dims,x0)
    print((d,n,p,loops),"f0",orc.cost(Q,x0),"gpu",got["f"],got["grad_norm"],got["iterations"],got["hvps"],got["status"],"| oracle",ref["f"],ref["grad_norm"],ref["iterations"],ref["hvps"],ref["status"], "sec",got["seconds"])
