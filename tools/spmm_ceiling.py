"""Where the LightGCN SpMM's time goes: the same kernels with the gathered rows confined to a table that fits one
XCD's L2 (gather ceiling of the cache hierarchy), and the user-row / item-row halves on their own."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gdmcf_amd
from gdmcf_amd import _lib, data
from gdmcf_amd.lightgcn import spmm_plan
lib = _lib.load(); dev = "cuda:0"
shape = sys.argv[1] if len(sys.argv) > 1 else "yelp"
cfg = data.SHAPES[shape]
indptr, indices, I = data.synth_csr(shape, seed=0)
U = cfg["n_users"]
users = np.repeat(np.arange(U), np.diff(indptr))
m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": indices}, U, I, 3, 64, device=dev).to(dev)
aptr, col, val = m.norm_adj_csr
N = U + I
X = torch.randn(N, 64, device=dev); Y = torch.empty_like(X)
arr = (ctypes.c_void_p * 1)()


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6


def make_runner(ptr_np, col_t, rows=None):
    """plan over the rows `rows` (None = all) of the CSR (ptr_np, col_t, val)"""
    if rows is not None:
        p = np.zeros(len(ptr_np), np.int64)  # same row numbering; the other rows become empty
        deg = np.diff(ptr_np); keep = np.zeros(len(deg), bool); keep[rows] = True
        # a plan is a list of (beg, end, row): build it on the kept rows only
        plan = spmm_plan(ptr_np)
        sel = keep[plan["vrow"]]
        ns = int(sel[:plan["n_short"]].sum())
        pl = {k: torch.from_numpy(plan[k][sel]).to(dev) for k in ("vbeg", "vend", "vrow", "vslot")}
        pl["lrow"], pl["lptr"] = torch.from_numpy(plan["lrow"]).to(dev), torch.from_numpy(plan["lptr"]).to(dev)
        pl["n_short"] = ns
    else:
        plan = spmm_plan(ptr_np)
        pl = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in plan.items()}
    part = torch.empty(max(plan["n_slots"], 1), 64, device=dev)
    nv, ns, nl = pl["vrow"].numel(), pl["n_short"], pl["lrow"].numel()

    def run():
        _lib.check(lib.gdmcf_spmm_csr_f32(pl["vbeg"].data_ptr(), pl["vend"].data_ptr(), pl["vrow"].data_ptr(), pl["vslot"].data_ptr(), nv, ns,
            pl["lrow"].data_ptr() if nl else None, pl["lptr"].data_ptr() if nl else None, nl, col_t.data_ptr(), val.data_ptr(), N,
            X.data_ptr(), 64, 64, Y.data_ptr(), 64, part.data_ptr(), arr, 0, 64, 1.0, 0.0, _lib.stream_ptr()))
    return run


ptr_np = aptr.cpu().numpy()
nnz = col.numel()
print(f"{shape}: N={N} nnz={nnz} gathered={nnz * 256 / 1e6:.0f} MB algorithmic={m.algorithmic_bytes() / 1e6:.1f} MB")
t = timeit(make_runner(ptr_np, col)); print(f"all rows, real columns          {t:7.1f} us   gather {nnz * 256 / t / 1e6:.2f} TB/s")
for span in (2048, 8192, 32768):
    c2 = (col % span).contiguous()
    t = timeit(make_runner(ptr_np, c2)); print(f"all rows, columns mod {span:6d}  ({span * 256 / 1e6:5.1f} MB table) {t:7.1f} us   gather {nnz * 256 / t / 1e6:.2f} TB/s")
ur, ir = np.arange(U), np.arange(U, N)
for name, rows in (("user rows", ur), ("item rows", ir)):
    g = int(np.diff(ptr_np)[rows].sum())
    t = timeit(make_runner(ptr_np, col, rows)); print(f"{name}, real columns         {t:7.1f} us   gather {g * 256 / t / 1e6:.2f} TB/s ({g} nnz)")
    c2 = (col % 2048).contiguous()
    t = timeit(make_runner(ptr_np, c2, rows)); print(f"{name}, columns mod 2048     {t:7.1f} us   gather {g * 256 / t / 1e6:.2f} TB/s")
