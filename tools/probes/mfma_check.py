import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, velesdb_amd as va
from oracle import pyoracle as po
va.set_sweep_engine(1)
rng=np.random.default_rng(0)
for metric,pm in ((va.DistanceMetric.Cosine,po.COSINE),(va.DistanceMetric.DotProduct,po.DOT)):
    for n,d in ((5000,768),(3000,100),(777,3),(2000,33),(4096,256),(1500,1024)):
        rows=rng.standard_normal((n,d)).astype(np.float32); 
        ix=va.HnswIndex(d,metric); ix.upload(np.arange(n),rows)
        for nq,k in ((1,10),(17,5),(40,10),(3,64)):
            Q=rng.standard_normal((nq,d)).astype(np.float32)
            gi,gs,gc=ix.search_batch_brute_force(Q,k)
            ei,es=po.scan_topk(pm,rows,Q,min(k,n),po.MODE_M,nthreads=4)
            ok=np.array_equal(gi[:,:ei.shape[1]],ei) and np.array_equal(gs[:,:es.shape[1]].view(np.uint32),es.view(np.uint32))
            if not ok:
                bad=np.argwhere(gs[:,:es.shape[1]].view(np.uint32)!=es.view(np.uint32))
                print("MISMATCH",metric,n,d,nq,k,len(bad), gs[0,:3],es[0,:3], gi[0,:3], ei[0,:3]); 
            else: print("ok",int(metric),n,d,nq,k)
        ix.close()
