"""Data formats of the reference (reference data_utils.py:164-226) and a device-resident batch provider.

`data_load` reads the reference's `{train,valid,test}_list.npy` files (int arrays [nnz, 2] of (uid, iid))
into scipy CSR float64 matrices with the reference's conventions (n_user / n_item from the TRAIN list only,
duplicate pairs summed by scipy's constructor).  `DataDiffusion` is the reference's dataset class.
`DeviceCSR` / `DeviceBatchLoader` keep the interaction matrix on the GPU as CSR and densify one batch at a
time in a HIP kernel, instead of materialising the dense [n_user, n_item] host matrix (main.py:143,148)
and copying 55 MB per batch over PCIe.
"""
import numpy as np
import scipy.sparse as sp
import torch
from torch.utils.data import Dataset

from . import _lib


def data_load(train_path, valid_path, test_path):
    train_list = np.load(train_path, allow_pickle=True)
    valid_list = np.load(valid_path, allow_pickle=True)
    test_list = np.load(test_path, allow_pickle=True)
    n_user = int(train_list[:, 0].max()) + 1
    n_item = int(train_list[:, 1].max()) + 1
    print(f"user num: {n_user}")
    print(f"item num: {n_item}")

    def csr(pairs):
        return sp.csr_matrix((np.ones_like(pairs[:, 0]), (pairs[:, 0], pairs[:, 1])), dtype="float64",
                             shape=(n_user, n_item))

    return csr(train_list), csr(valid_list), csr(test_list), n_user, n_item


class DataDiffusion(Dataset):
    def __init__(self, data):
        self.data = data

    def __getitem__(self, index):
        return self.data[index], index

    def __len__(self):
        return len(self.data)


class DeviceCSR:
    """A scipy CSR interaction matrix resident in HBM (indptr int64, indices int32, values float32)."""

    def __init__(self, csr, device="cuda"):
        csr = sp.csr_matrix(csr)
        csr.sum_duplicates()
        self.shape = csr.shape
        self.device = torch.device(device)
        self.indptr = torch.from_numpy(csr.indptr.astype(np.int64)).to(self.device)
        self.indices = torch.from_numpy(csr.indices.astype(np.int32)).to(self.device)
        vals = csr.data.astype(np.float32)
        self.values = None if np.all(vals == 1.0) else torch.from_numpy(vals).to(self.device)

    def rows(self, row_ids, out=None):
        """Dense float32 [len(row_ids), n_items] block on the device (row_ids: int64 tensor or None = first rows)."""
        ids = row_ids.to(device=self.device, dtype=torch.int64).contiguous()
        B, I = ids.numel(), self.shape[1]
        if out is None:
            out = torch.empty(B, I, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().gdmcf_densify_rows_f32(self.indptr.data_ptr(), self.indices.data_ptr(),
                                                      _lib.ptr(self.values), ids.data_ptr(), B, I, out.data_ptr(),
                                                      out.stride(0), _lib.stream_ptr()))
        return out

    def batch(self, row_ids):
        """The same rows, left sparse (see CsrBatch)."""
        return CsrBatch(self, row_ids)


class CsrBatch:
    """A batch of interaction rows that stays sparse: rows `row_ids` of a DeviceCSR.  Hand it to
    `GaussianDiffusion.training_losses` in place of the dense `[B, n_items]` tensor of the reference's loop
    (main.py:343-346): the input builder then reads the few hundred bytes of CSR per row instead of a dense row, and the loss
    takes its {0,1} target from bitmaps (gdmcf_dnn_prep_input_csr_f32 / gdmcf_linear_loss_fwd_bits_f32; same arithmetic,
    bit-identical losses and gradients).  Configurations that need the dense row (F.normalize, the eps target, the
    one-hot variants, values other than 1) densify it themselves (`dense()`)."""

    def __init__(self, csr, row_ids):
        self.csr = csr
        self.row_ids = row_ids.to(device=csr.device, dtype=torch.int64).contiguous()
        self.shape = (self.row_ids.numel(), csr.shape[1])
        self.device = csr.device
        self.is_cuda = csr.device.type == "cuda"

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 2

    def dense(self, out=None):
        return self.csr.rows(self.row_ids, out=out)


class DeviceBatchLoader:
    """Iterates (dense_batch_on_device, index) like DataLoader(DataDiffusion(dense), batch_size, shuffle,
    drop_last) does in the reference (main.py:153-156), without host densification or PCIe traffic."""

    def __init__(self, csr, batch_size, shuffle=False, drop_last=False, device="cuda", generator=None, n_users=None,
                 sparse=False, ids_only=False):
        """sparse=True yields the rows as `CsrBatch` (never densified: `training_losses` takes them as they are);
        ids_only=True yields (None, index) for a consumer that gathers the rows itself (graph.GraphedTrainStep)."""
        self.csr = csr if isinstance(csr, DeviceCSR) else DeviceCSR(csr, device)
        self.batch_size, self.shuffle, self.drop_last, self.generator = batch_size, shuffle, drop_last, generator
        self.sparse, self.ids_only = sparse, ids_only
        self.n = self.csr.shape[0] if n_users is None else min(n_users, self.csr.shape[0])

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else -(-self.n // self.batch_size)

    def __iter__(self):
        order = torch.randperm(self.n, generator=self.generator) if self.shuffle else torch.arange(self.n)
        for lo in range(0, self.n, self.batch_size):
            idx = order[lo:lo + self.batch_size]
            if self.drop_last and idx.numel() < self.batch_size:
                return
            if self.ids_only:
                yield None, idx
            else:
                yield (self.csr.batch(idx) if self.sparse else self.csr.rows(idx)), idx
