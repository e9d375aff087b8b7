"""The training / evaluation loop surface of the reference driver (reference main.py:267-351), restated
on top of the HIP path.  Host glue only: batches in, metrics out.

* `train_one_epoch`  = main.py:327-351 (the body that is dead code behind a stray `continue` in the
  shipped file): zero_grad -> training_losses -> mean -> backward -> step per batch.
* `evaluate`         = main.py:267-310: p_sample -> history mask -> top-N -> computeTopNAccuracy.
  The reference moves a dense [B, I] row block to the device per batch and masks with
  `prediction[his_data.nonzero()] = -inf`; here the history stays CSR and the mask is applied inside the
  top-k kernel.
"""
import numpy as np
import torch

from . import evaluate_utils
from .data_utils import DeviceBatchLoader, DeviceCSR
from .evaluate_utils import masked_topk
from .parallel import DataParallelStep


def dense_rows(csr, rows, device):
    """float32 dense [len(rows), n_items] block of a scipy CSR matrix, on `device` (what DataDiffusion +
    DataLoader deliver in the reference, data_utils.py:216-226)."""
    return torch.from_numpy(np.asarray(csr[rows].todense(), dtype=np.float32)).to(device)


def train_one_epoch(diffusion, model, optimizer, train_csr, batch_size, device, reweight=True, shuffle=True,
                    drop_last=True, generator=None, step=None, sparse=False, graph_step=None):
    """One pass over the users of `train_csr`.  Returns (sum of batch losses, number of batches), the two
    numbers the reference prints per epoch (main.py:377).  sparse=True hands `training_losses` the rows as a CsrBatch
    (bit-identical, no densify launch); graph_step = a graph.GraphedTrainStep over the same matrix replays the whole
    step from one hipGraph per batch (drop_last is then forced: the captured batch size is fixed)."""
    model.train()
    loader = DeviceBatchLoader(train_csr, batch_size, shuffle=shuffle, drop_last=drop_last or graph_step is not None,
                               device=device, generator=generator, sparse=sparse, ids_only=graph_step is not None)
    step = step or (None if graph_step is not None else DataParallelStep(diffusion, model, optimizer))
    total, count = None, 0
    with_index = bool(getattr(diffusion, "indexIn", False))  # embedding backbones need the users' ids (main.py:346)
    for batch, index in loader:
        if graph_step is not None:
            loss = graph_step(index)
        else:
            loss = step(batch, reweight, index=index) if with_index else step(batch, reweight)
        total = loss if total is None else total + loss
        count += 1
    return (float(total) if total is not None else 0.0), count


@torch.no_grad()
def evaluate(diffusion, model, data_csr, data_te, mask_his, topN, sampling_steps, sampling_noise, batch_size,
             device):
    """Precision / Recall / NDCG / MRR @topN exactly as reference main.py:267-310.

    data_csr: rows fed to p_sample (the reference feeds the training rows); data_te: ground-truth CSR;
    mask_his: CSR of interactions to exclude from the ranking."""
    model.eval()
    n = mask_his.shape[0]
    predict_items = []
    dcsr = data_csr if isinstance(data_csr, DeviceCSR) else DeviceCSR(data_csr, device)
    for lo in range(0, n, batch_size):
        rows = np.arange(lo, min(lo + batch_size, n))
        batch = dcsr.rows(torch.from_numpy(rows))
        kw = dict(index=torch.from_numpy(rows)) if getattr(diffusion, "indexIn", False) else {}
        prediction = diffusion.p_sample(model, batch, sampling_steps, sampling_noise, **kw)
        indptr, cols = evaluate_utils.csr_rows_to_device(mask_his, rows, device)
        predict_items.append(masked_topk(prediction, topN[-1], indptr, cols))
    # the ranked lists never leave the device; the metric terms per user are computed there too
    return evaluate_utils.computeTopNAccuracy_device(data_te, torch.cat(predict_items), topN)
