"""Reference point for the evaluation tail: the reference's history mask + torch.topk(prediction, 100) (main.py:296-301)
next to gdmcf_topk_masked_f32, on 400 rows of the Yelp shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gdmcf_amd  # noqa: E402

dev = "cuda:0"
B, I, k = 400, 34395, 100
g = torch.Generator(device=dev).manual_seed(0)
pred = torch.randn(B, I, device=dev, generator=g)
his = (torch.rand(B, I, device=dev, generator=g) < 0.00075)
rows, cols = his.nonzero(as_tuple=True)
csr = his.float().to_sparse_csr()
indptr, indices = csr.crow_indices(), csr.col_indices().to(torch.int32)


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


def torch_path():
    p = pred.clone()
    p[rows, cols] = -float("inf")
    return torch.topk(p, k).indices


t_ref = timeit(torch_path)
t_ours = timeit(lambda: gdmcf_amd.masked_topk(pred, k, indptr, indices))
same = sum(set(a.tolist()) == set(b.tolist()) for a, b in zip(torch_path().cpu(), gdmcf_amd.masked_topk(pred, k, indptr, indices).cpu()))
print(f"mask + top-{k} of {B} x {I}: torch (clone + index_put + topk) {t_ref:.3f} ms,  gdmcf_topk_masked_f32 {t_ours:.3f} ms;  "
      f"index sets equal on {same}/{B} rows")
