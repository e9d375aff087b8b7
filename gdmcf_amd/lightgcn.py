"""LightGCN embedding propagation with the reference's interface (reference lightGCN.py:129-203).

`get_A_tilda` builds the symmetric-normalised bipartite adjacency D^-1/2 A D^-1/2 once on the host
(float32, as the reference) and keeps it on the GPU as CSR; `propagate_through_layers` runs the
n_layers SpMMs in the HIP kernel gdmcf_spmm_csr_f32 with the layer mean fused as a running sum.
Only the forward propagation is on the hot path; BPR training is a
"next" row (8f3), so E0 gradients are not produced here.
"""
import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

from . import _lib


def normalized_bipartite_csr(users, items, n_users, n_items):
    """CSR (indptr int64, indices int32, data float32) of A~ = D^-1/2 [[0,R],[R^T,0]] D^-1/2,
    d = (rowsum + 1e-9)^-1/2 in float32, duplicates collapsed (reference :146-164)."""
    users = np.asarray(users, dtype=np.int64)
    items = np.asarray(items, dtype=np.int64)
    R = sp.coo_matrix((np.ones(len(users), np.float32), (users, items)), shape=(n_users, n_items)).tocsr()
    R.data[:] = 1.0
    A = sp.bmat([[None, R], [R.T.tocsr(), None]], format="csr", dtype=np.float32)
    A.sort_indices()
    N = n_users + n_items
    rowsum = np.asarray(A.sum(1), dtype=np.float32).flatten()
    d = np.power(rowsum + np.float32(1e-9), np.float32(-0.5)).astype(np.float32)
    d[np.isinf(d)] = 0.0
    rows = np.repeat(np.arange(N), np.diff(A.indptr))
    data = ((d[rows] * A.data).astype(np.float32) * d[A.indices]).astype(np.float32)
    return A.indptr.astype(np.int64), A.indices.astype(np.int32), data


class LightGCN(nn.Module):
    def __init__(self, data, n_users, n_items, n_layers, latent_dim, device="cuda"):
        """data: DataFrame/dict with `user_id_idx` / `item_id_idx` columns (reference :147)."""
        super().__init__()
        self.data = data
        self.n_users, self.n_items = n_users, n_items
        self.n_layers, self.latent_dim = n_layers, latent_dim
        self._device = torch.device(device)
        self.init_embedding()
        self.norm_adj_csr = self.get_A_tilda()

    def init_embedding(self):
        self.E0 = nn.Embedding(self.n_users + self.n_items, self.latent_dim)
        nn.init.xavier_uniform_(self.E0.weight)
        self.E0.weight = nn.Parameter(self.E0.weight)

    def get_A_tilda(self):
        indptr, indices, vals = normalized_bipartite_csr(np.asarray(self.data["user_id_idx"]),
                                                         np.asarray(self.data["item_id_idx"]), self.n_users,
                                                         self.n_items)
        dev = self._device
        return (torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev), torch.from_numpy(vals).to(dev))

    @torch.no_grad()
    def propagate_through_layers(self, return_layers=False):
        lib = _lib.load()
        E0 = self.E0.weight.detach()
        _lib.require_gpu(E0, "LightGCN.E0")
        indptr, indices, vals = self.norm_adj_csr
        N, d = E0.shape
        st = _lib.stream_ptr()
        acc = E0.clone()
        cur = E0.contiguous()
        layers = []
        bufs = [torch.empty_like(acc), torch.empty_like(acc)]
        for layer in range(self.n_layers):
            nxt = bufs[layer & 1]
            _lib.check(lib.gdmcf_spmm_csr_f32(indptr.data_ptr(), indices.data_ptr(), vals.data_ptr(), N, cur.data_ptr(),
                                              cur.stride(0), d, nxt.data_ptr(), nxt.stride(0), acc.data_ptr(),
                                              acc.stride(0), st))
            if return_layers:
                layers.append(nxt.clone())
            cur = nxt
        mean = torch.empty_like(acc)
        _lib.check(lib.gdmcf_scale_f32(acc.data_ptr(), acc.numel(), 1.0 / (self.n_layers + 1), mean.data_ptr(), st))
        final_user, final_item = torch.split(mean, [self.n_users, self.n_items])
        init_user, init_item = torch.split(E0, [self.n_users, self.n_items])
        if return_layers:
            return final_user, final_item, init_user, init_item, layers
        return final_user, final_item, init_user, init_item

    def forward(self, users, pos_items, neg_items):
        fu, fi, iu, ii = self.propagate_through_layers()
        return fu[users], fi[pos_items], fi[neg_items], iu[users], ii[pos_items], ii[neg_items]
