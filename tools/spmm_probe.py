"""Splits the LightGCN SpMM of the Yelp-shape graph into its user-row and item-row halves and times each."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gdmcf_amd
from gdmcf_amd import _lib, data
lib = _lib.load(); dev = "cuda:0"
shape = sys.argv[1] if len(sys.argv) > 1 else "yelp"
cfg = data.SHAPES[shape]
indptr, indices, I = data.synth_csr(shape, seed=0)
U = cfg["n_users"]
users = np.repeat(np.arange(U), np.diff(indptr))
m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": indices}, U, I, 3, 64, device=dev).to(dev)
pl = m._plan; _, col, val = m.norm_adj_csr
X = torch.randn(U + I, 64, device=dev); Y = torch.empty_like(X)
nv, ns = pl["vrow"].numel(), pl["n_short"]
print("virtual rows", nv, "short", ns, "split rows", pl["lrow"].numel(), "nnz", m.nnz)
arr = (ctypes.c_void_p * 1)()
def run(v0, v1, nshort, use_long):
    off = lambda t, i, sz: t.data_ptr() + i * sz
    nl = pl["lrow"].numel() if use_long else 0
    _lib.check(lib.gdmcf_spmm_csr_f32(off(pl["vbeg"], v0, 8), off(pl["vend"], v0, 8), off(pl["vrow"], v0, 4), off(pl["vslot"], v0, 4), v1 - v0, nshort,
        pl["lrow"].data_ptr() if nl else None, pl["lptr"].data_ptr() if nl else None, nl, col.data_ptr(), val.data_ptr(), U + I,
        X.data_ptr(), 64, 64, Y.data_ptr(), 64, m._partial.data_ptr(), arr, 0, 64, 1.0, 0.0, _lib.stream_ptr()))
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
print("all         %.1f us" % timeit(lambda: run(0, nv, ns, True)))
print("short rows  %.1f us" % timeit(lambda: run(0, ns, ns, False)))
print("long rows   %.1f us" % timeit(lambda: run(ns, nv, 0, True)))
