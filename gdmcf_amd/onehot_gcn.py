"""GCN backbone `DNNOneHotEmbeddingGCN` (reference models/DNN.py:1077-1103, :1105-1327) -- the one the shipped YAML
selects; third slice of SURVEY 8 f1, args.noise_type == 0.

PARITY UNPINNED: `GCNConv` is torch_geometric==2.5.3 (requirements.txt:53), absent from the reference tree and from this
image, and the reference class constructs itself with `.cuda()`; the oracle restates GCNConv from its published semantics
(oracle/gdmcf_oracle.py:gcn_conv) and this module is tested against that restatement.

The backbone is DNNOneHotEmbedding whose user-side vector hc = [h, h_U, embedding_user(index)] is replaced by
`hc * sumW + gcn([hc; items], edges)[:B] * (1 - sumW)` before the cosine scores (:1277-1289).  The edges run from user b to
item i only, GCNConv aggregates at the TARGET of an edge, and only the user rows of its output are used: a user node
receives nothing but its own self loop (normalisation 1), so those rows are a plain perceptron of hc,
    conv1: hc @ W1^T + b1  ->  ReLU (the LeakyReLU after it is the identity)  ->  conv2: @ W2^T + b2,
independent of `graph` (tests/test_oracle_golden.py pins this on the full-graph restatement).  That is what runs here:
two more dense layers on the C-ABI GEMMs; the ReLU and the scalar blend act on [B, 3*hid] / [B, 512] tensors and use
torch's elementwise kernels on the device.
"""
import math

import torch
import torch.nn as nn

from . import _lib
from .onehot import _ceil64
from .onehot_embedding import DNNOneHotEmbedding, OneHotEmbeddingEngine


class OneHotGCNEngine(OneHotEmbeddingEngine):
    def buffers(self, B, device):
        b = super().buffers(B, device)
        if hasattr(b, "ublend"):
            return b
        m = self.model
        f32 = dict(dtype=torch.float32, device=device)
        b.ublend = torch.zeros(B, b.ucat.stride(0), **f32)
        b.gcn_hid = m.gcn_model.conv1.lin.weight.shape[0] if m.gcn_layers == 2 else 0  # (no gcn_model at all for 0 layers)
        if m.gcn_layers == 2:
            b.z1 = torch.zeros(B, _ceil64(b.gcn_hid), **f32)
            b.dz1 = torch.zeros_like(b.z1)
        b.z2 = torch.zeros(B, b.ucat.stride(0), **f32)
        b.dtmp = torch.zeros(B, b.ucat.stride(0), **f32)
        ws = b.ws_bytes
        for conv in self._convs():
            w = conv.lin.weight
            ws = max(ws, self.lib.gdmcf_linear_ws_bytes(B, w.shape[0], w.shape[1]), self.lib.gdmcf_linear_ws_bytes(B, w.shape[1], w.shape[0]))
        if ws > b.ws_bytes:
            b.ws_bytes = int(ws)
            b.ws = torch.empty(ws, dtype=torch.uint8, device=device)
        return b

    def _convs(self):
        if self.model.gcn_layers == 0:
            return []
        g = self.model.gcn_model
        return [g.conv1] + ([g.conv2] if self.model.gcn_layers == 2 else [])

    def _linear(self, bufs, B, A, conv, out):
        w, bias = conv.lin.weight, conv.bias
        _lib.check(self.lib.gdmcf_linear_fwd_f32(A.data_ptr(), A.stride(0), w.data_ptr(), w.stride(0), bias.data_ptr(), 0, B,
                                                 w.shape[0], w.shape[1], out.data_ptr(), out.stride(0), bufs.ws.data_ptr(),
                                                 bufs.ws_bytes, _lib.stream_ptr()))

    def _user_vector(self, bufs, B):
        m, D = self.model, bufs.D
        if m.gcn_layers == 0:
            return bufs.ucat  # hc * sumW + hc * (1 - sumW)
        convs = self._convs()
        if m.gcn_layers == 2:
            self._linear(bufs, B, bufs.ucat, convs[0], bufs.z1)
            bufs.z1[:, : bufs.gcn_hid].relu_()
            self._linear(bufs, B, bufs.z1, convs[1], bufs.z2)
        else:
            self._linear(bufs, B, bufs.ucat, convs[0], bufs.z2)
        s = m.sumW.detach()
        torch.add(bufs.ucat[:, :D] * s, bufs.z2[:, :D] * (1 - s), out=bufs.ublend[:, :D])  # reference :1288
        return bufs.ublend

    def _user_vector_backward(self, bufs, B):
        m, D, lib, st = self.model, bufs.D, self.lib, _lib.stream_ptr()
        if m.gcn_layers == 0:
            return {m.sumW: torch.zeros_like(m.sumW)}
        convs = self._convs()
        s = m.sumW.detach()
        du = bufs.du[:, :D]
        grads = {m.sumW: (du * (bufs.ucat[:, :D] - bufs.z2[:, :D])).sum().reshape(())}
        bufs.dtmp[:, :D] = du * (1 - s)  # d z2
        du.mul_(s)                        # d hc, direct term

        def layer_backward(conv, dz, A, dA):
            w, bias = conv.lin.weight, conv.bias
            N, K = w.shape
            dW, db = torch.empty_like(w), torch.empty_like(bias)
            _lib.check(lib.gdmcf_linear_bwd_weight_f32(dz.data_ptr(), dz.stride(0), A.data_ptr(), A.stride(0), None, 0, B, N, K,
                                                       dW.data_ptr(), dW.stride(0), db.data_ptr(), 0, st))
            _lib.check(lib.gdmcf_linear_bwd_input_f32(dz.data_ptr(), dz.stride(0), w.data_ptr(), w.stride(0), None, A.data_ptr(),
                                                      A.stride(0), 0, B, N, K, dA.data_ptr(), dA.stride(0), bufs.ws.data_ptr(),
                                                      bufs.ws_bytes, st))
            grads[w], grads[bias] = dW, db

        if m.gcn_layers == 2:
            layer_backward(convs[1], bufs.dtmp, bufs.z1, bufs.dz1)
            bufs.dz1[:, : bufs.gcn_hid].mul_(bufs.z1[:, : bufs.gcn_hid] > 0)  # ReLU'
            layer_backward(convs[0], bufs.dz1, bufs.ucat, bufs.z2)  # z2 is free now: receives d hc through the perceptron
        else:
            layer_backward(convs[0], bufs.dtmp, bufs.ucat, bufs.z2)
        du.add_(bufs.z2[:, :D])
        return grads

    def _train_backward(self, gloss):
        m = self.model
        base = super()._train_backward(gloss)  # gradients in DNNOneHotEmbedding's parameter order
        names = [k for k, _ in m.named_parameters() if not (k.startswith("gcn_model") or k == "sumW")]
        by_param = {id(dict(m.named_parameters())[k]): g for k, g in zip(names, base)}
        for p, g in self._extra_grads.items():
            if self.grad_sink is not None:
                self.grad_sink(p, g)
                g = None
            by_param[id(p)] = g
        return [by_param.get(id(p)) for p in m.parameters()]


class _GCNConvParams(nn.Module):
    """Parameter container with torch_geometric 2.5 GCNConv's names and initialisation: `lin.weight` [out, in] (glorot
    uniform, no bias in `lin`) and `bias` (zeros)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.lin = nn.Linear(cin, cout, bias=False)
        self.bias = nn.Parameter(torch.zeros(cout))
        a = math.sqrt(6.0 / (cin + cout))
        self.lin.weight.data.uniform_(-a, a)


class _LayerGCNParams(nn.Module):
    def __init__(self, cin, hidden, cout, layers):
        super().__init__()
        if layers == 1:
            self.conv1 = _GCNConvParams(cin, cout)
        else:
            self.conv1 = _GCNConvParams(cin, hidden)
            self.conv2 = _GCNConvParams(hidden, cout)


class DNNOneHotEmbeddingGCN(DNNOneHotEmbedding):
    """Drop-in for the reference DNNOneHotEmbeddingGCN (main.py:243-246), args.noise_type == 0; `args.gcnLayerNum` (0, 1,
    2; parse_args_util.py:24 defaults to 2) may also be given as `gcn_layers=`."""

    def __init__(self, in_dims, out_dims, emb_size, time_type="cat", norm=False, dropout=0.5, item_num=2810, user_num=5949,
                 args=None, gcn_layers=None, gemm_dtype="f32"):
        self.args = args
        if gcn_layers is None:
            gcn_layers = int(getattr(args, "gcnLayerNum", 2))
        if getattr(args, "noise_type", 0) != 0:
            raise NotImplementedError("DNNOneHotEmbeddingGCN: only args.noise_type == 0 (the ablations 1 / 2 are not built)")
        if gcn_layers not in (0, 1, 2):
            raise ValueError("gcnLayerNum must be 0, 1 or 2")
        self._gcn_layers_pending = gcn_layers
        super().__init__(in_dims, out_dims, emb_size, time_type=time_type, norm=norm, dropout=dropout, item_num=item_num,
                         user_num=user_num, gemm_dtype=gemm_dtype)
        self.gcn_layers = gcn_layers
        d = self.embedding_item.weight.shape[1]
        if gcn_layers > 0:
            self.gcn_model = _LayerGCNParams(d, 512, d, gcn_layers)  # :1150-1155 (hidden_dim = 512)
        self.sumW = nn.Parameter(torch.tensor(1.0))

    @property
    def engine(self):
        if self._engine is None:
            self._engine = OneHotGCNEngine(self)
        return self._engine
