"""LightGCN SpMM, one layer: second-generation bundled schedule vs the first-generation kernels, with the real columns
and with the gathered rows confined to a table that fits one XCD's L2 (the cache hierarchy's gather ceiling).
    python tools/spmm_probe2.py [yelp|stress] [gen]      env: GDMCF_SPMM_WAVES / _SMAX / _PIECE"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
shape = sys.argv[1] if len(sys.argv) > 1 else "yelp"
if len(sys.argv) > 2:
    os.environ["GDMCF_SPMM_GEN"] = sys.argv[2]
import gdmcf_amd
from gdmcf_amd import _lib, data
lib = _lib.load(); dev = "cuda:0"
cfg = data.SHAPES[shape]
indptr, indices, I = data.synth_csr(shape, seed=0)
U = cfg["n_users"]
users = np.repeat(np.arange(U), np.diff(indptr))
t0 = time.time()
m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": indices}, U, I, 1, 64, device=dev).to(dev)
print(f"{shape}: graph + schedule built in {time.time() - t0:.2f} s; generation {'3 (streamed)' if m._streamed else '2 (bundled)' if m._bundled else '1'}; "
      f"waves {m._plan.get('n_waves')}, pieces {m._plan.get('n_pieces')}, bundles {m._plan.get('n_bundles')}, units "
      f"{m._plan.get('n_units')}, entries {m._plan.get('n_entries')}, cut rows "
      f"{m._plan['crow'].numel() if (m._bundled or m._streamed) else m._plan['lrow'].numel()}")
nnz, alg = m.nnz, m.algorithmic_bytes()
X = torch.randn(U + I, 64, device=dev)


def timeit(n=50):
    """(wall us per call with the host enqueueing back to back, HIP-event us of the launches themselves)"""
    import ctypes
    with torch.no_grad():
        for _ in range(5): m._propagate(X)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): m._propagate(X)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) / n * 1e6
        lib.gdmcf_prof_enable(1)
        for _ in range(10): m._propagate(X)
        torch.cuda.synchronize()
        cap = 4096
        tags, ms, work = (ctypes.c_int * cap)(), (ctypes.c_float * cap)(), (ctypes.c_double * cap)()
        k = lib.gdmcf_prof_collect(cap, tags, ms, work)
        lib.gdmcf_prof_enable(0)
        ev = sorted(float(ms[i]) for i in range(k) if tags[i] == 8)
    return ev[len(ev) // 2] * 1e3, wall


t, wall = timeit()
print(f"real columns            {t:8.1f} us/layer (wall {wall:.1f})  algorithmic {alg / t / 1e6:.2f} TB/s = {alg / t / 8e6:.3f} of 8 TB/s; gathered {nnz * 256 / t / 1e6:.2f} TB/s")
if m._streamed:
    col = m._plan["cw"].view(-1, 2)[:, 0]
else:
    col = m.norm_adj_csr[1]
keep = col.clone()
for span in (2048, 8192):
    col.copy_(keep % span)
    t, wall = timeit()
    print(f"columns mod {span:5d}       {t:8.1f} us/layer (wall {wall:.1f})  gathered {nnz * 256 / t / 1e6:.2f} TB/s")
col.copy_(keep)
if m._bundled:
    # where does the time go?  (1) (col, val) reads confined to 32 KB (L2 hits instead of HBM streaming) on top of the
    # confined columns; (2) additionally every row written to the same 2048 rows of Y
    pl = m._plan
    col.copy_(keep % 2048)
    sb, lb = pl["sbeg"].clone(), pl["lbeg"].clone()
    pl["sbeg"].copy_(sb % 4096); pl["lbeg"].copy_(lb % 4096)
    t, wall = timeit()
    print(f"+ (col,val) from 32 KB    {t:8.1f} us/layer (wall {wall:.1f})  gathered {nnz * 256 / t / 1e6:.2f} TB/s")
    sr, lr = pl["srow"].clone(), pl["lrow"].clone()
    pl["srow"].copy_(torch.where(sr >= 0, sr % 2048, sr)); pl["lrow"].copy_(lr % 2048)
    t, wall = timeit()
    print(f"+ Y rows mod 2048         {t:8.1f} us/layer (wall {wall:.1f})  gathered {nnz * 256 / t / 1e6:.2f} TB/s")
    pl["sbeg"].copy_(sb); pl["lbeg"].copy_(lb); pl["srow"].copy_(sr); pl["lrow"].copy_(lr); col.copy_(keep)
if m._bundled:
    # pieces alone / bundles alone, real columns: do the range-cut pieces run at L2 speed?
    pl = m._plan
    wd = pl["wdesc"].view(-1, 4)
    keepw = wd.clone()
    nz_p = int(pl["llen"].sum()); nz_b = nnz - nz_p
    wd[:, 3] = wd[:, 2]
    t, wall = timeit()
    print(f"pieces only ({nz_p} nnz)  {t:8.1f} us (wall {wall:.1f})  gathered {nz_p * 256 / t / 1e6:.2f} TB/s")
    wd.copy_(keepw); wd[:, 1] = wd[:, 0]
    t, wall = timeit()
    print(f"bundles only ({nz_b} nnz) {t:8.1f} us (wall {wall:.1f})  gathered {nz_b * 256 / t / 1e6:.2f} TB/s")
    wd.copy_(keepw)
