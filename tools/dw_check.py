"""Weight-gradient product through the C ABI (gdmcf_linear_bwd_weight_f32) against torch, with / without bias sums and row
scales -- isolates the DR kernel's epilogue and bias paths.   python tools/dw_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdmcf_amd import _lib
lib = _lib.load(); dev = "cuda:0"
torch.manual_seed(0)
for (B, N, K) in [(400, 34395, 1000), (400, 1000, 34405), (400, 94949, 1000), (400, 1000, 94959), (256, 5000, 777)]:
    ldz, lda = (N + 63) // 64 * 64, (K + 63) // 64 * 64
    dZ = torch.randn(B, ldz, device=dev); A = torch.randn(B, lda, device=dev); rs = torch.rand(B, device=dev) + 0.5
    ref = (dZ[:, :N].double().t() @ A[:, :K].double())
    for use_db, use_rs in [(0, 0), (1, 0), (1, 1)]:
        dW = torch.full((N, K), float("nan"), device=dev); db = torch.full((N,), float("nan"), device=dev)
        _lib.check(lib.gdmcf_linear_bwd_weight_f32(dZ.data_ptr(), ldz, A.data_ptr(), lda, rs.data_ptr() if use_rs else None, 0, B, N, K,
                                                   dW.data_ptr(), K, db.data_ptr() if use_db else None, 0, _lib.stream_ptr()))
        torch.cuda.synchronize()
        err = float((dW.double() - ref).abs().max() / ref.abs().max())
        bad = int(torch.isnan(dW).sum())
        msg = f"B={B} N={N} K={K} db={use_db} rs={use_rs}: dW max err {err:.2e}, nan {bad}"
        if use_db:
            dref = (dZ[:, :N].double() * (rs.double()[:, None] if use_rs else 1.0)).sum(0)
            msg += f", db err {float((db.double() - dref).abs().max() / dref.abs().max()):.2e} nan {int(torch.isnan(db).sum())}"
        print(msg, flush=True)
        if err > 1e-3:
            bad_el = ((dW.double() - ref).abs() > 1e-3 * ref.abs().max())
            rows = bad_el.any(1).nonzero().flatten(); cols = bad_el.any(0).nonzero().flatten()
            print(f"    wrong elements {int(bad_el.sum())} in {rows.numel()} rows x {cols.numel()} cols; rows {rows[:12].tolist()} ... cols {cols[:12].tolist()} .. {cols[-4:].tolist()}")
            r0 = int(rows[0]); print("    row", r0, "got", dW[r0, :6].tolist(), "ref", ref[r0, :6].tolist())
