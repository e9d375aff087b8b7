"""Evaluation helpers with the reference's interface (reference evaluate_utils.py:6-69) plus the
device-side history-mask + top-k used by main.py:296-301."""
import math

import numpy as np

import torch

from . import _lib


def masked_topk(prediction, k, mask_indptr=None, mask_indices=None, return_values=False):
    """Top-k item indices per row after setting the user's history to -inf (reference main.py:299-301).

    prediction: float32 [B, I] on the GPU (not modified).  The history mask is CSR: `mask_indptr`
    int64 [B+1], `mask_indices` int32 column ids.  Scores come back in descending order; ties are
    broken by the LOWEST item index (torch.topk leaves tie order unspecified)."""
    _lib.require_gpu(prediction, "prediction")
    lib = _lib.load()
    pred = prediction if (prediction.dtype == torch.float32 and prediction.stride(-1) == 1) else \
        prediction.float().contiguous()
    B, I = pred.shape
    if not 1 <= k <= I:
        raise AssertionError("selected index k out of range")
    dev = pred.device
    ip = ix = None
    if mask_indptr is not None:
        ip = mask_indptr.to(device=dev, dtype=torch.int64).contiguous()
        ix = mask_indices.to(device=dev, dtype=torch.int32).contiguous()
        assert ip.numel() == B + 1, "mask_indptr must have B+1 entries"
    idx = torch.empty(B, k, dtype=torch.int64, device=dev)
    val = torch.empty(B, k, dtype=torch.float32, device=dev) if return_values else None
    _lib.check(lib.gdmcf_topk_masked_f32(pred.data_ptr(), pred.stride(0), B, I, _lib.ptr(ip), _lib.ptr(ix), k,
                                         idx.data_ptr(), _lib.ptr(val), _lib.stream_ptr()))
    return (val, idx) if return_values else idx


def csr_rows_to_device(csr, rows, device):
    """(indptr int64, indices int32) device tensors for `csr[rows]` (a scipy CSR history matrix)."""
    sub = csr[rows]
    return (torch.from_numpy(sub.indptr.astype("int64")).to(device),
            torch.from_numpy(sub.indices.astype("int32")).to(device))


def computeTopNAccuracy(GroundTruth, predictedIndices, topN):
    """Precision / Recall / NDCG / MRR @N with the reference's exact conventions: users with an
    empty ground truth add nothing to the sums but still count in the divisor; 4-decimal rounding."""
    precision, recall, NDCG, MRR = [], [], [], []
    n_users = len(predictedIndices)
    for N in topN:
        sum_p = sum_r = sum_n = sum_m = 0
        for i in range(n_users):
            gt = GroundTruth[i]
            if len(gt) == 0:
                continue
            gt_set = set(gt)
            hits, dcg, idcg, mrr, left, first = 0, 0.0, 0.0, 0.0, len(gt), True
            pred = predictedIndices[i]
            for j in range(N):
                if pred[j] in gt_set:
                    dcg += 1.0 / math.log2(j + 2)
                    if first:
                        mrr = 1.0 / (j + 1.0)
                        first = False
                    hits += 1
                if left > 0:
                    idcg += 1.0 / math.log2(j + 2)
                    left -= 1
            sum_p += hits / N
            sum_r += hits / len(gt)
            sum_n += (dcg / idcg) if idcg != 0 else 0
            sum_m += mrr
        precision.append(round(sum_p / n_users, 4))
        recall.append(round(sum_r / n_users, 4))
        NDCG.append(round(sum_n / n_users, 4))
        MRR.append(round(sum_m / n_users, 4))
    return precision, recall, NDCG, MRR


def computeTopNAccuracy_device(ground_truth_csr, predicted_idx, topN):
    """computeTopNAccuracy with the per-user work on the MI355X (gdmcf_topn_metrics_f64): `ground_truth_csr` is a
    scipy CSR (one row per predicted user), `predicted_idx` an int64 device tensor [U, >= max(topN)] as masked_topk
    returns it.  The per-user terms are bit-identical to the Python loop; they are added up in user order on the
    host, so the rounded results equal computeTopNAccuracy's (the reference's) exactly."""
    import ctypes
    from . import _lib
    _lib.require_gpu(predicted_idx, "predicted indices")
    topN = [int(n) for n in topN]
    if topN != sorted(set(topN)) or not topN:
        raise ValueError("topN must be ascending and non-empty")
    gt = ground_truth_csr.tocsr().astype(np.float32, copy=True)
    gt.eliminate_zeros()
    gt.sort_indices()
    U = gt.shape[0]
    pred = predicted_idx if predicted_idx.dtype == torch.int64 else predicted_idx.to(torch.int64)
    if pred.stride(-1) != 1:
        pred = pred.contiguous()
    assert pred.shape[0] == U and pred.shape[1] >= topN[-1], "one prediction row per ground-truth row, >= max(topN) items"
    dev = pred.device
    indptr = torch.from_numpy(gt.indptr.astype(np.int64)).to(dev)
    indices = torch.from_numpy(gt.indices.astype(np.int32)).to(dev)
    out = torch.empty(U, len(topN), 4, dtype=torch.float64, device=dev)
    tn = (ctypes.c_int * len(topN))(*topN)
    _lib.check(_lib.load().gdmcf_topn_metrics_f64(pred.data_ptr(), pred.stride(0), U, indptr.data_ptr(), indices.data_ptr(),
                                                  tn, len(topN), out.data_ptr(), _lib.stream_ptr()))
    terms = out.cpu().numpy()
    sums = np.add.accumulate(terms, axis=0)[-1]  # sequential float64 additions in user order, as the reference's loop
    res = [[round(float(sums[k, m]) / U, 4) for k in range(len(topN))] for m in range(4)]
    return res[0], res[1], res[2], res[3]


def print_results(loss, valid_result, test_result):
    """output the evaluation results."""
    if loss is not None:
        print("[Train]: loss: {:.4f}".format(loss))
    for tag, res in (("Valid", valid_result), ("Test", test_result)):
        if res is not None:
            print("[{}]: Precision: {} Recall: {} NDCG: {} MRR: {}".format(
                tag, *["-".join(str(x) for x in res[i]) for i in range(4)]))
