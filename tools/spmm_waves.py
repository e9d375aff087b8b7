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
# warm / cold columns as the plan sees them: the most gathered SPMM_HOT_MB of the table are warm
from gdmcf_amd.lightgcn import SPMM_HOT_MB
n_hot = int(SPMM_HOT_MB * 1e6 // 256)
thresh = np.partition(cnt, len(cnt) - n_hot)[len(cnt) - n_hot]
cold_col = cnt < max(thresh, 1)
ent_per_unit = (groups * UN * G).astype(np.int64)
feat = np.zeros((nw, 6))
for w in range(nw):
    sb, nb_, u0, u1 = wd[w]
    pos = sb * 64
    for u in range(u0, u1):
        n = ent_per_unit[u]
        e = cw[pos:pos + n]
        real = e[:, 1] != 0
        ncold = int(cold_col[e[real, 0]].sum()); nreal = int(real.sum())
        if is_piece[u]:
            feat[w, 0] += groups[u]; feat[w, 3] += 1
        elif ncold * 2 > nreal:
            feat[w, 2] += groups[u]; feat[w, 5] += 1
        else:
            feat[w, 1] += groups[u]; feat[w, 4] += 1
        pos += n
y = end - first
A = np.column_stack([feat, np.ones(nw)])
coef, *_ = np.linalg.lstsq(A, y, rcond=None)
pred = A @ coef
print("time per wave ~ %.3f us/piece-group + %.3f us/warm-bundle-group + %.3f us/cold-bundle-group + %.3f us/piece + %.3f us/warm bundle + "
      "%.3f us/cold bundle + %.2f us;  residual rms %.2f us (of mean %.1f)" % (*coef, np.sqrt(np.mean((y - pred) ** 2)), y.mean()))
print("groups per wave: pieces", q(feat[:, 0]), "\n                 warm  ", q(feat[:, 1]), "\n                 cold  ", q(feat[:, 2]))
print("units per wave:  pieces", q(feat[:, 3]), "\n                 warm  ", q(feat[:, 4]), "\n                 cold  ", q(feat[:, 5]))
for c in range(8):
    sl = slice(c * wpc, (c + 1) * wpc)
    print(f"class {c}: groups pieces {feat[sl, 0].sum():7.0f} warm {feat[sl, 1].sum():7.0f} cold {feat[sl, 2].sum():7.0f}; units {feat[sl, 3].sum():6.0f} {feat[sl, 4].sum():6.0f} {feat[sl, 5].sum():6.0f}; predicted wave time {pred[sl].mean():5.1f}, measured {y[sl].mean():5.1f}")
order = np.argsort(-end)[:12]
print("slowest waves:  wave  class  end    first  pieces_done  predicted(end-first)  measured  groups(p/w/c)  units(p/w/c)  block")
for w_ in order:
    print(f"   {w_:5d} {w_ // wpc:3d} {end[w_]:7.1f} {first[w_]:6.1f} {pieces[w_]:7.1f}   {pred[w_]:6.1f} {y[w_]:6.1f}   "
          f"{feat[w_, 0]:.0f}/{feat[w_, 1]:.0f}/{feat[w_, 2]:.0f}   {feat[w_, 3]:.0f}/{feat[w_, 4]:.0f}/{feat[w_, 5]:.0f}   {int(st[w_, 7])}")
res = y - pred
print("residual (measured - predicted) by position of the wave in its class: first 8th %.2f, last 8th %.2f; by wave-in-block: %s"
      % (np.mean([res[c * wpc:c * wpc + wpc // 8].mean() for c in range(8)]), np.mean([res[(c + 1) * wpc - wpc // 8:(c + 1) * wpc].mean() for c in range(8)]),
         [round(float(res[i::4].mean()), 2) for i in range(4)]))
