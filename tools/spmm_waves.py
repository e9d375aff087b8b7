"""Per-wave timeline of the streamed SpMM kernel (gdmcf_debug_spmm_stamps): when do waves start, reach their first gather,
finish their pieces, finish -- per XCD class."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GDMCF_SPMM_GEN", "3")
import numpy as np, torch
import gdmcf_amd
from gdmcf_amd import _lib, data
lib = _lib.load(); dev = "cuda:0"
shape = sys.argv[1] if len(sys.argv) > 1 else "yelp"
cfg = data.SHAPES[shape]
indptr, indices, I = data.synth_csr(shape, seed=0)
U = cfg["n_users"]
users = np.repeat(np.arange(U), np.diff(indptr))
m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": indices}, U, I, 1, 64, device=dev).to(dev)
X = torch.randn(U + I, 64, device=dev)
with torch.no_grad():
    for _ in range(5): m._propagate(X)
    torch.cuda.synchronize()
    nw = m._plan["n_waves"]
    _lib.check(lib.gdmcf_debug_spmm_stamps(nw, None))
    m._propagate(X)
    buf = (ctypes.c_longlong * (nw * 8))()
    _lib.check(lib.gdmcf_debug_spmm_stamps(nw, buf))
st = np.frombuffer(buf, dtype=np.int64).reshape(nw, 8).astype(np.float64)
t0 = st[:, 0].min()
us = lambda v: (v - t0) / 100.0  # 100 MHz ticks -> microseconds
start, first, pieces, end = us(st[:, 0]), us(st[:, 1]), us(st[:, 2]), us(st[:, 3])
print(f"{shape}: {nw} waves; kernel span {end.max():.1f} us")
q = lambda a: "min %6.1f  p10 %6.1f  median %6.1f  p90 %6.1f  max %6.1f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
print("wave start             ", q(start))
print("first gather - start   ", q(first - start))
print("pieces done - first    ", q(pieces - first))
print("end - pieces done      ", q(end - pieces))
print("wave end               ", q(end))
print("units per wave         ", q(st[:, 4]), " batches per wave", q(st[:, 5]))
wpc = nw // 8
for c in range(8):
    sl = slice(c * wpc, (c + 1) * wpc)
    print(f"class {c} (XCC {int(st[sl, 6][0])}): start {np.median(start[sl]):5.1f}  first {np.median(first[sl]):5.1f}  pieces {np.median(pieces[sl]):5.1f}  end median {np.median(end[sl]):5.1f} max {end[sl].max():5.1f}")
# ---- what does a wave's time depend on?  least squares of (end - first gather) on its steps by kind and its units ----
pl = m._plan
DW, UN, G = pl["DW"], pl["UN"], pl["G"]
ud = pl["ud"].cpu().numpy().reshape(-1, DW)
wd = pl["wdesc"].cpu().numpy().reshape(-1, 4)
hdr = ud[:, 0]
groups = (hdr & 0x7FFFFFFF).astype(np.float64)
is_piece = hdr < 0
# a bundle is "cold" when the plan marked it so: recompute from the columns it gathers (most gathered 2 MB = warm)
cw = pl["cw"].cpu().numpy().reshape(-1, 2)
cnt = np.bincount(cw[:, 0], minlength=U + I)
feat = np.zeros((nw, 4))
for w in range(nw):
    u0, u1 = wd[w, 2], wd[w, 3]
    g_ = groups[u0:u1]; p_ = is_piece[u0:u1]
    feat[w] = [g_[p_].sum(), g_[~p_].sum(), p_.sum(), (~p_).sum()]
y = end - first
A = np.column_stack([feat, np.ones(nw)])
coef, *_ = np.linalg.lstsq(A, y, rcond=None)
pred = A @ coef
print("time per wave ~ %.3f us/piece-group + %.3f us/bundle-group + %.3f us/piece + %.3f us/bundle + %.2f us;  residual rms %.2f us (of mean %.1f)"
      % (*coef, np.sqrt(np.mean((y - pred) ** 2)), y.mean()))
print("groups per wave: pieces", q(feat[:, 0]), " bundles", q(feat[:, 1]))
