"""LightGCN embedding propagation with the reference's interface (reference lightGCN.py:129-203).

`get_A_tilda` builds the symmetric-normalised bipartite adjacency D^-1/2 A D^-1/2 once on the host
(float32, as the reference), keeps it on the GPU as CSR and builds the execution plan that splits hub
rows into virtual rows; `propagate_through_layers` runs the n_layers SpMMs in the HIP kernel
gdmcf_spmm_csr_f32 with the layer mean fused into the last layer's epilogue.
Only the forward propagation is on the hot path; BPR training is a
"next" row (8f3), so E0 gradients are not produced here.
"""
import ctypes

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

from . import _lib

import os

SPMM_CHUNK = int(os.environ.get("GDMCF_SPMM_CHUNK", "256"))  # max nonzeros per virtual row


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


SPMM_SHORT = int(os.environ.get("GDMCF_SPMM_SHORT", "48"))  # rows up to this many nonzeros run four-to-a-wave


def spmm_plan(indptr, chunk=SPMM_CHUNK, short=SPMM_SHORT, d=64):
    """Virtual-row plan for gdmcf_spmm_csr_f32 (see include/gdmcf_hip.h).  Rows with <= `short` nonzeros come
    first, whole (they run 64/(d/4) to a wave); every other row is cut into pieces of <= chunk nonzeros; rows cut
    into more than one piece get partial slots + an entry in lrow/lptr."""
    indptr = np.asarray(indptr, dtype=np.int64)
    deg = np.diff(indptr)
    lpr = d // 4
    if d % 4 != 0 or lpr not in (2, 4, 8, 16, 32, 64):
        short = -1  # the short-row kernel needs power-of-two lane groups
    is_short = deg <= short
    srows = np.nonzero(is_short)[0]
    lrows_all = np.nonzero(~is_short)[0]
    ldeg = deg[lrows_all]
    pieces = np.maximum(1, -(-ldeg // chunk)).astype(np.int64)
    n_rest = int(pieces.sum())
    rrow = np.repeat(lrows_all, pieces)
    first = np.concatenate([[0], np.cumsum(pieces)[:-1]]) if len(pieces) else np.zeros(0, np.int64)
    k = np.arange(n_rest, dtype=np.int64) - np.repeat(first, pieces)  # piece index inside its row
    rbeg = indptr[rrow] + k * chunk
    rend = np.minimum(rbeg + chunk, indptr[rrow + 1])
    cut = np.repeat(pieces > 1, pieces)
    rslot = np.full(n_rest, -1, dtype=np.int32)
    rslot[cut] = np.arange(int(cut.sum()), dtype=np.int32)
    lrow = lrows_all[pieces > 1].astype(np.int32)
    lptr = np.concatenate([[0], np.cumsum(pieces[pieces > 1])]).astype(np.int32)
    return dict(vbeg=np.concatenate([indptr[srows], rbeg]).astype(np.int64),
                vend=np.concatenate([indptr[srows + 1], rend]).astype(np.int64),
                vrow=np.concatenate([srows, rrow]).astype(np.int32),
                vslot=np.concatenate([np.full(len(srows), -1, np.int32), rslot]).astype(np.int32),
                lrow=lrow, lptr=lptr, n_slots=int(cut.sum()), n_short=int(len(srows)))


class LightGCN(nn.Module):
    def __init__(self, data, n_users, n_items, n_layers, latent_dim, device="cuda", shard_rows=False, group=None):
        """data: DataFrame/dict with `user_id_idx` / `item_id_idx` columns (reference :147).

        shard_rows (new, multi-GPU; the reference is single-device): every rank of `group` keeps only a contiguous
        block of ceil(N/world) rows of the normalised adjacency.  A layer is then a local SpMM on the full embedding
        table of the previous layer plus an all-gather of the row blocks (N*d*4/world bytes per rank per layer over
        xGMI); the backward pass all-reduces the cotangent first -- propagation is linear, so sum_r A~^l g_r =
        A~^l sum_r g_r -- which makes E0.grad the SUM over ranks already (do not all-reduce it again)."""
        super().__init__()
        import torch.distributed as dist
        self._group = group
        self._world = dist.get_world_size(group) if (shard_rows and dist.is_initialized()) else 1
        self._rank = dist.get_rank(group) if self._world > 1 else 0
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
        N = self.n_users + self.n_items
        rpr = -(-N // self._world)  # rows per rank
        r0 = min(self._rank * rpr, N)
        r1 = min(r0 + rpr, N)
        self._rows = (r0, r1, rpr)
        if self._world > 1:
            lo, hi = int(indptr[r0]), int(indptr[r1])
            indptr, indices, vals = indptr[r0:r1 + 1] - indptr[r0], indices[lo:hi], vals[lo:hi]
        plan = spmm_plan(indptr, d=self.latent_dim)
        self._plan = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in plan.items()}
        self._partial = torch.empty(max(plan["n_slots"], 1), self.latent_dim, dtype=torch.float32, device=dev)
        self.nnz = int(indices.size)
        return (torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev), torch.from_numpy(vals).to(dev))

    def algorithmic_bytes(self):
        """Compulsory HBM bytes of one layer: CSR once, X once, Y once."""
        N, d = self.n_users + self.n_items, self.latent_dim
        rows = self._rows[1] - self._rows[0]  # this rank's block: CSR once, X once, its rows of Y once
        return self.nnz * 8.0 + (rows + 1) * 8.0 + (N + rows) * d * 4.0

    @torch.no_grad()
    def _propagate(self, X, return_layers=False):
        """mean_l (A~^l X), l = 0..n_layers, through gdmcf_spmm_csr_f32 (layer mean fused into the last SpMM)."""
        lib = _lib.load()
        _lib.require_gpu(X, "LightGCN embeddings")
        _, indices, vals = self.norm_adj_csr
        pl = self._plan
        N, d = X.shape
        st = _lib.stream_ptr()
        cur = X.contiguous()
        layers = [cur]
        nv, nl = pl["vrow"].numel(), pl["lrow"].numel()
        r0, r1, rpr = self._rows
        sharded = self._world > 1
        for layer in range(self.n_layers):
            last = (layer == self.n_layers - 1) and not return_layers
            out = torch.zeros(rpr, d, dtype=cur.dtype, device=cur.device) if sharded else torch.empty_like(cur)
            adds = layers if last else []
            # the fused layer mean reads this rank's rows of the earlier layers
            arr = (ctypes.c_void_p * max(len(adds), 1))(*[a.data_ptr() + r0 * a.stride(0) * 4 for a in adds])
            if r1 > r0:
                _lib.check(lib.gdmcf_spmm_csr_f32(
                    pl["vbeg"].data_ptr(), pl["vend"].data_ptr(), pl["vrow"].data_ptr(), pl["vslot"].data_ptr(), nv,
                    pl["n_short"], _lib.ptr(pl["lrow"]) if nl else None,
                    _lib.ptr(pl["lptr"]) if nl else None, nl, indices.data_ptr(), vals.data_ptr(), r1 - r0,
                    cur.data_ptr(), cur.stride(0), d, out.data_ptr(), out.stride(0), self._partial.data_ptr(), arr,
                    len(adds), cur.stride(0), 1.0 / (self.n_layers + 1) if last else 1.0, self.algorithmic_bytes(), st))
            if sharded:
                from .parallel import all_gather_rows
                out = all_gather_rows(out, N, self._group)
            layers.append(out)
            cur = out
        if return_layers:
            mean = torch.stack(layers).sum(0) / (self.n_layers + 1)  # test/debug path only
            return mean, layers[1:]
        return (cur if self.n_layers > 0 else X), None

    def propagate_through_layers(self, return_layers=False):
        """(final_user, final_item, initial_user, initial_item) as reference lightGCN.py:180-194.  Differentiable
        w.r.t. E0 when autograd is on (the backward is the same propagation: A~ is symmetric)."""
        E0 = self.E0.weight
        if return_layers:
            mean, layers = self._propagate(E0.detach(), return_layers=True)
        elif torch.is_grad_enabled() and E0.requires_grad:
            mean, layers = _PropagateMean.apply(self, E0), None
        else:
            mean, layers = self._propagate(E0.detach())[0], None
        final_user, final_item = torch.split(mean, [self.n_users, self.n_items])
        init_user, init_item = torch.split(E0 if layers is None else E0.detach(), [self.n_users, self.n_items])
        if return_layers:
            return final_user, final_item, init_user, init_item, layers
        return final_user, final_item, init_user, init_item

    def forward(self, users, pos_items, neg_items):
        fu, fi, iu, ii = self.propagate_through_layers()
        return fu[users], fi[pos_items], fi[neg_items], iu[users], ii[pos_items], ii[neg_items]


class _PropagateMean(torch.autograd.Function):
    """mean over layers of A~^l E0.  d(mean)/d(E0) applied to a cotangent G is mean_l (A~^l)^T G = mean_l A~^l G
    because the normalised adjacency is symmetric (reference lightGCN.py:149-164): the backward pass is the
    forward kernel run on the gradient."""

    @staticmethod
    def forward(ctx, module, E0):
        ctx.module = module
        return module._propagate(E0.detach())[0]

    @staticmethod
    def backward(ctx, g):
        m = ctx.module
        g = g.contiguous()
        if m._world > 1:  # row-sharded: propagate the SUM of all ranks' cotangents (linearity), see LightGCN.__init__
            from .parallel import _all_reduce
            g = g.clone()
            _all_reduce(g, m._group)
        return None, m._propagate(g)[0]


def bpr_loss(users, users_emb, pos_emb, neg_emb, userEmb0, posEmb0, negEmb0):
    """(mf_loss, reg_loss) of reference lightGCN.py:207-219."""
    reg_loss = (1 / 2) * (userEmb0.norm().pow(2) + posEmb0.norm().pow(2) + negEmb0.norm().pow(2)) / float(len(users))
    pos_scores = torch.sum(torch.mul(users_emb, pos_emb), dim=1)
    neg_scores = torch.sum(torch.mul(users_emb, neg_emb), dim=1)
    loss = torch.mean(torch.nn.functional.softplus(neg_scores - pos_scores))
    return loss, reg_loss


def sample_bpr_batch(indptr, indices, n_users, n_items, batch_size, rng):
    """(users, pos_items, neg_items) as the reference sampler (lightGCN.py:221-251, minus its stray
    pdb.set_trace): sorted users drawn without replacement (with replacement if n_users < batch_size), one random
    interacted item and one random non-interacted item per user.  Vectorised numpy; `rng` = np.random.Generator."""
    if n_users < batch_size:
        users = np.sort(rng.integers(0, n_users, batch_size))
    else:
        users = np.sort(rng.choice(n_users, batch_size, replace=False))
    deg = indptr[users + 1] - indptr[users]
    if np.any(deg == 0):
        raise ValueError("sample_bpr_batch: every sampled user needs at least one interaction")
    pos = indices[indptr[users] + (rng.random(batch_size) * deg).astype(np.int64)]
    neg = rng.integers(0, n_items, batch_size)
    for _ in range(64):  # rejection: resample the (few) negatives that hit an interacted item
        bad = np.array([n in indices[indptr[u]:indptr[u + 1]] for u, n in zip(users, neg)]) if batch_size <= 4096 else \
            np.zeros(batch_size, bool)
        if not bad.any():
            break
        neg[bad] = rng.integers(0, n_items, int(bad.sum()))
    return users.astype(np.int64), pos.astype(np.int64), neg.astype(np.int64)
