"""Reference point for the LightGCN propagation: torch.sparse.mm (rocSPARSE) with the normalised adjacency as COO (what
the reference's lightGCN.py:176,185 runs) and as CSR, next to gdmcf_spmm_csr_f32, on the Yelp-shape synthetic graph."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gdmcf_amd  # noqa: E402
from gdmcf_amd import data  # noqa: E402

dev = "cuda:0"
cfg = data.SHAPES["yelp"]
indptr, indices, I = data.synth_csr("yelp", seed=0)
U = cfg["n_users"]
users = np.repeat(np.arange(U), np.diff(indptr))
torch.manual_seed(0)
m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": indices}, U, I, 3, 64, device=dev).to(dev)
ip, idx, val = m.norm_adj_csr
N = U + I
csr = torch.sparse_csr_tensor(ip, idx.to(torch.int64), val, size=(N, N))
coo = csr.to_sparse_coo().coalesce()
X = m.E0.weight.detach()


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


t_coo = timeit(lambda: torch.sparse.mm(coo, X))
t_csr = timeit(lambda: torch.sparse.mm(csr, X))
with torch.no_grad():
    t_ours = timeit(lambda: m.propagate_through_layers()) / 3
    ref = torch.sparse.mm(csr, X)
    got = m._propagate(X, return_layers=True)[1][0]
print(f"nnz {m.nnz}, N {N}, d 64:  torch.sparse.mm COO {t_coo:.3f} ms/layer,  CSR {t_csr:.3f} ms/layer,  "
      f"gdmcf_spmm_csr_f32 {t_ours:.3f} ms/layer (incl. fused layer mean);  max |diff| vs torch CSR {float((ref - got).abs().max()):.2e}")
