"""`indexIn` backbone `DNNOneHotEmbedding` (reference models/DNN.py:510-682) on the HIP path -- second slice of SURVEY 8 f1.

Same two input branches as DNNOneHot; instead of `out_layers` (which exist, are initialised and never applied, :582-592
/ :658-662) the concatenation u = [h, h_U, embedding_user(index)] is scored against every row of `embedding_item` by
cosine similarity (:655, :667-682).  With RCloss the NT-Xent term between the two hidden activations (:479-508,
:641-643) is returned too; GaussianDiffusionDiscrete adds 0.1 x it to every row's loss (:952-953).

Device work: the scores are `gdmcf_linear_loss_fwd_f32` (training) / `gdmcf_linear_fwd_f32` (evaluation) on the
row-normalised operands u/|u| and V/|v| (`gdmcf_row_norms_f32` + `gdmcf_rowscale_f32`), their gradients are the usual
input / weight gradient GEMMs followed by the backward of the normalisation (`gdmcf_normalize_rows_bwd_f32`); the user
rows move with `gdmcf_gather_rows_f32` / `gdmcf_scatter_add_rows_f32`; tanh' of the hidden activations takes the NT-Xent
gradient as an addend (`gdmcf_tanh_bwd_f32`).  The NT-Xent term itself is a softmax over a [B, B] matrix of the two
[B, hid] activations -- 0.01 % of the step's arithmetic -- and is evaluated with the reference's own torch expressions on
the device (forward and gradient).
"""
import torch
import torch.nn as nn

from . import _lib
from .onehot import DNNOneHot, OneHotEngine, _ceil64


def nt_xent_loss(z1, z2, temperature=0.1, eps=1e-5):
    """reference models/DNN.py:479-508 (its `loss2`)."""
    n = z1.size(0)
    sim = torch.softmax(torch.mm(z1, z2.t()) / temperature, dim=-1)
    mask = torch.eye(n, device=z1.device).bool()
    negatives = sim.masked_select(~mask).view(n, -1)
    return -torch.log((torch.diag(sim) + eps) / negatives.sum(dim=1)).mean()


class OneHotEmbeddingEngine(OneHotEngine):
    def buffers(self, B, device):
        b = super().buffers(B, device)
        if hasattr(b, "ucat"):
            return b
        m, lib = self.model, self.lib
        f32 = dict(dtype=torch.float32, device=device)
        b.h12 = b.h1 + b.h2
        b.eu = m.embedding_user.weight.shape[1]
        b.D = b.h12 + b.eu
        if m.embedding_item.weight.shape != (self.I, b.D):
            raise RuntimeError("gdmcf_amd.DNNOneHotEmbedding: embedding_item must be [n_items, h1 + h2 + user width]")
        ldD = _ceil64(b.D)
        b.ucat = torch.zeros(B, ldD, **f32)   # [h | h_U | user row]; the branches write their column ranges directly
        b.hcat = b.ucat
        b.dhcat = torch.zeros(B, ldD, **f32)  # d(pre-activation) behind ucat[:, :h12]
        b.uhat = torch.zeros(B, ldD, **f32)
        b.du = torch.zeros(B, ldD, **f32)
        b.hs = torch.zeros(B, ldD, **f32)
        b.rn_u = torch.zeros(B, **f32)
        b.rn_v = torch.zeros(self.I, **f32)
        b.Vhat = torch.zeros(self.I, ldD, **f32)
        ws = max(b.ws_bytes, lib.gdmcf_linear_ws_bytes(B, self.I, b.D))
        if ws > b.ws_bytes:
            b.ws_bytes = int(ws)
            b.ws = torch.empty(ws, dtype=torch.uint8, device=device)
        return b

    def _scores_operands(self, bufs, br1, br2, B, index):
        """ucat = [h, h_U, embedding_user(index)], then the row-normalised operands uhat, Vhat of the cosine scores."""
        lib, st, m = self.lib, _lib.stream_ptr(), self.model
        self._hidden(bufs, br1, br2, [None], B)
        Wu, V = m.embedding_user.weight, m.embedding_item.weight
        ld = bufs.ucat.stride(0)
        _lib.check(lib.gdmcf_gather_rows_f32(Wu.data_ptr(), Wu.stride(0), index.data_ptr(), B, bufs.eu,
                                             bufs.ucat.data_ptr() + 4 * bufs.h12, ld, st))
        u = self._user_vector(bufs, B)  # what is scored against the items: ucat itself, or a subclass's function of it
        _lib.check(lib.gdmcf_row_norms_f32(u.data_ptr(), u.stride(0), B, bufs.D, None, bufs.rn_u.data_ptr(), st))
        _lib.check(lib.gdmcf_rowscale_f32(u.data_ptr(), u.stride(0), bufs.rn_u.data_ptr(), B, bufs.D, bufs.uhat.data_ptr(),
                                          bufs.uhat.stride(0), st))
        # V / |v| only changes with the item table (every optimiser step while training; never during evaluation, where
        # the reverse loop calls the model T times per batch): rebuilt when the parameter's version counter moved
        key = (V.data_ptr(), V._version)
        if getattr(bufs, "vhat_key", None) != key:
            _lib.check(lib.gdmcf_row_norms_f32(V.data_ptr(), V.stride(0), self.I, bufs.D, None, bufs.rn_v.data_ptr(), st))
            _lib.check(lib.gdmcf_rowscale_f32(V.data_ptr(), V.stride(0), bufs.rn_v.data_ptr(), self.I, bufs.D,
                                              bufs.Vhat.data_ptr(), bufs.Vhat.stride(0), st))
            bufs.vhat_key = key

    def _user_vector(self, bufs, B):
        return bufs.ucat

    def _user_vector_backward(self, bufs, B):
        """bufs.du holds the gradient w.r.t. the scored user vector; turn it into the gradient w.r.t. ucat (in place) and
        return {parameter: gradient} of whatever lies between the two."""
        return {}

    @staticmethod
    def _index_on(index, device, B):
        if index is None:
            raise RuntimeError("gdmcf_amd.DNNOneHotEmbedding needs the users' ids (`index`)")
        index = index.to(device=device, dtype=torch.int64).contiguous()
        if index.shape != (B,):
            raise RuntimeError("gdmcf_amd.DNNOneHotEmbedding: `index` must hold one user id per row")
        return index

    def _train_forward(self, spec):
        B, dev = spec["x_start"].shape[0], spec["x_start"].device
        br1, br2, out = self._chains()
        bufs = self.buffers(B, dev)
        self.version += 1
        index = self._index_on(spec["index"], dev, B)
        x0, target, alpha, rowdiv, keep = self._train_inputs(spec, bufs)
        self._scores_operands(bufs, br1, br2, B, index)
        loss = self._loss_layer(spec, bufs, B, bufs.uhat.data_ptr(), bufs.uhat.stride(0), bufs.Vhat.data_ptr(),
                                bufs.Vhat.stride(0), None, self.I, bufs.D, target, alpha, rowdiv)
        # NT-Xent term between the two hidden activations (reference torch expressions, [B, B] work)
        with torch.enable_grad():
            h = bufs.ucat[:, : bufs.h1].detach().clone().requires_grad_(True)
            hU = bufs.ucat[:, bufs.h1: bufs.h12].detach().clone().requires_grad_(True)
            closs = nt_xent_loss(h, hU)
            dh, dhU = torch.autograd.grad(closs, (h, hU))
        self.last_closs = closs.detach()
        self._saved = dict(B=B, bufs=bufs, chains=(br1, br2, out), index=index, closs_grad=torch.cat([dh, dhU], dim=1).contiguous(),
                           keepalive=(x0, keep, target, alpha, rowdiv, spec["pt"]))
        return loss + self.last_closs * 0.1  # reference :952-953 (after the history update and the division by pt)

    def _train_backward(self, gloss):
        """Gradients in model.parameters() order: emb_layer, in_layers, in_layers2, out_layers (None: never applied),
        embedding_item, embedding_user."""
        sv = self._saved
        if sv is None:
            raise RuntimeError("gdmcf_amd: train_backward without a preceding training_losses")
        lib, st, m = self.lib, _lib.stream_ptr(), self.model
        bufs, B, index = sv["bufs"], sv["B"], sv["index"]
        br1, br2, out = sv["chains"]
        rs = self._rowscale_of(bufs, gloss)
        # every row's loss carries + 0.1 * closs: d(total)/d(closs) = 0.1 * sum of the upstream row gradients
        if isinstance(gloss, float):
            gc = torch.full((1,), 0.1 * gloss * B, dtype=torch.float32, device=bufs.ucat.device)
        else:
            gc = (0.1 * gloss.sum()).to(torch.float32).reshape(1)
        V, Wu = m.embedding_item.weight, m.embedding_user.weight
        # scores = uhat @ Vhat^T: gradient w.r.t. Vhat, then through V / |v|
        dV, _ = self._weight_grad(bufs, B, V, None, bufs.diff.data_ptr(), bufs.ldi, rs, bufs.uhat.data_ptr(), bufs.uhat.stride(0))
        _lib.check(lib.gdmcf_normalize_rows_bwd_f32(dV.data_ptr(), dV.stride(0), bufs.Vhat.data_ptr(), bufs.Vhat.stride(0),
                                                    bufs.rn_v.data_ptr(), self.I, bufs.D, dV.data_ptr(), dV.stride(0), st))
        # ... w.r.t. uhat, then through u / |u|
        self._input_grad(bufs, B, bufs.Vhat.data_ptr(), bufs.Vhat.stride(0), self.I, bufs.D, bufs.diff.data_ptr(), bufs.ldi, rs,
                         bufs.ucat.data_ptr(), bufs.ucat.stride(0), 0, bufs.du.data_ptr(), bufs.du.stride(0))
        _lib.check(lib.gdmcf_normalize_rows_bwd_f32(bufs.du.data_ptr(), bufs.du.stride(0), bufs.uhat.data_ptr(),
                                                    bufs.uhat.stride(0), bufs.rn_u.data_ptr(), B, bufs.D, bufs.du.data_ptr(),
                                                    bufs.du.stride(0), st))
        self._extra_grads = self._user_vector_backward(bufs, B)
        # user rows: scatter into the dense table gradient (torch.optim.AdamW on nn.Embedding sees a dense gradient too)
        dWu = torch.zeros_like(Wu)
        _lib.check(lib.gdmcf_scatter_add_rows_f32(bufs.du.data_ptr() + 4 * bufs.h12, bufs.du.stride(0), index.data_ptr(), B,
                                                  bufs.eu, dWu.data_ptr(), dWu.stride(0), st))
        # hidden activations: + NT-Xent gradient, times tanh'
        cg = sv["closs_grad"]
        _lib.check(lib.gdmcf_tanh_bwd_f32(bufs.du.data_ptr(), bufs.du.stride(0), bufs.ucat.data_ptr(), bufs.ucat.stride(0),
                                          cg.data_ptr(), cg.stride(0), gc.data_ptr(), B, bufs.h12, bufs.dhcat.data_ptr(),
                                          bufs.dhcat.stride(0), st))
        res = self._branches_backward(bufs, B, br1, br2)
        res += [None, None] * len(out)
        if self.grad_sink is not None:
            self.grad_sink(V, dV)
            self.grad_sink(Wu, dWu)
            dV = dWu = None
        sv["keepalive"] = (sv["keepalive"], gc)
        return res + [dV, dWu]

    def forward_plain(self, x, timesteps, x_U, training, drop_mask=None, drop_mask_U=None, index=None, posterior=None):
        prev = self.lib.gdmcf_gemm_precision(self._precision())
        try:
            B, dev = x.shape[0], x.device
            br1, br2, _ = self._chains()
            bufs = self.buffers(B, dev)
            lib, st = self.lib, _lib.stream_ptr()
            self.version += 1
            self._saved = None
            index = self._index_on(index, dev, B)
            ts = timesteps.to(device=dev, dtype=torch.int64).contiguous()
            if x.dtype != torch.float32 or x.stride(-1) != 1:
                x = x.float().contiguous()
            xu = x_U.reshape(B, -1)
            if xu.shape[1] != 2 * self.I:
                raise RuntimeError("gdmcf_amd.DNNOneHotEmbedding: x_U must hold two columns per item")
            if xu.dtype != torch.float32 or xu.stride(-1) != 1:
                xu = xu.float().contiguous()
            keep = (self._prep(bufs, x, self.I, bufs.xin1, ts, None, None, None, drop_mask, training),
                    self._prep(bufs, xu, 2 * self.I, bufs.xin2, ts, None, None, None, drop_mask_U, training))
            self._scores_operands(bufs, br1, br2, B, index)
            res = self._last_layer(bufs.uhat.data_ptr(), bufs.uhat.stride(0), bufs.Vhat.data_ptr(), bufs.Vhat.stride(0), None,
                                   0, B, self.I, bufs.D, x, posterior, bufs, st)
            del keep
            return res
        finally:
            self.lib.gdmcf_gemm_precision(prev)


class DNNOneHotEmbedding(DNNOneHot):
    """Drop-in for the reference DNNOneHotEmbedding (models/DNN.py:510-682); main.py:239-242 builds it with
    `item_num=n_item, user_num=n_user` and sets `diffusion.indexIn = True`."""

    def __init__(self, in_dims, out_dims, emb_size, time_type="cat", norm=False, dropout=0.5, item_num=2810, user_num=5949,
                 gemm_dtype="f32"):
        self._defer_init = True
        super().__init__(in_dims, out_dims, emb_size, time_type=time_type, norm=norm, dropout=dropout, gemm_dtype=gemm_dtype)
        eu = self.in_layers[-1].out_features
        self.embedding_item = nn.Embedding(item_num, eu + eu + self.in_layers2[-1].out_features)
        self.embedding_user = nn.Embedding(user_num, eu)
        self.all_indices_item = torch.arange(item_num)
        self.all_indices_user = torch.arange(user_num)
        self._defer_init = False
        self.init_weights()
        self.lrelu = torch.nn.LeakyReLU(0.1)

    def init_weights(self):
        if getattr(self, "_defer_init", False):  # the reference draws once, after the embedding tables exist (:556)
            return
        super().init_weights()
        nn.init.xavier_uniform_(self.embedding_item.weight)
        nn.init.xavier_uniform_(self.embedding_user.weight)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = OneHotEmbeddingEngine(self)
        return self._engine

    @torch.no_grad()
    def load_lightgcn_embeddings(self, lightgcn, users=True, items=True):
        """Hand-off of the LightGCN propagation into the denoiser (the reference's script only saves
        final_user_Embed / final_item_Embed, lightGCN.py:305-323, and nothing reads them).  The propagated tables
        mean_l(A~^l E0) (HIP SpMM, gdmcf_amd.LightGCN.propagate_through_layers) initialise the tables this backbone
        conditions on (reference models/DNN.py:1148-1149, read at :1263-1265 / :1274):
          embedding_user.weight          <- final_user                       (the user row appended to [h, h_U], :1274)
          embedding_item.weight[:, -w:]  <- final_item, w = user width       (the columns that meet the user row in the
                                                                              cosine score, :1288-1289)
        so that score(u, i) contains <e_u, e_i> of the graph model from the first step on.  Needs latent_dim == the user
        embedding width (in_dims[-1] by construction, :1144).  Returns (final_user, final_item)."""
        fu, fi, _, _ = lightgcn.propagate_through_layers()
        wu = self.embedding_user.weight
        if users:
            if fu.shape != wu.shape:
                raise ValueError(f"LightGCN user table {tuple(fu.shape)} does not fit embedding_user {tuple(wu.shape)}")
            wu.copy_(fu)
        if items:
            wi = self.embedding_item.weight
            if fi.shape != (wi.shape[0], wu.shape[1]):
                raise ValueError(f"LightGCN item table {tuple(fi.shape)} does not fit the last {wu.shape[1]} columns of "
                                 f"embedding_item {tuple(wi.shape)}")
            wi[:, wi.shape[1] - wu.shape[1]:].copy_(fi)
        for p_ in (self.embedding_user.weight, self.embedding_item.weight):
            torch.autograd.graph.increment_version(p_)  # cached V/|v| and shadows are keyed on the version counter
        return fu, fi

    def forward(self, x, timesteps, x_U, index=None, graph=None, RCloss=False, drop_mask=None, drop_mask_U=None, posterior=None):
        """model(x_t, t, x_tU, index=..., graph=...) of the reference's evaluation path (`graph` is accepted and, as in the
        reference, unused).  Training goes through GaussianDiffusionDiscrete.training_losses.  `posterior`: see
        DNNOneHot.forward (the reverse loop's posterior mean fused into the score GEMM)."""
        _lib.require_gpu(x, "DNNOneHotEmbedding input")
        if RCloss or (torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters())):
            raise RuntimeError("gdmcf_amd.DNNOneHotEmbedding: the plain forward is not differentiable and carries no "
                               "NT-Xent term; train through GaussianDiffusionDiscrete.training_losses")
        return self.engine.forward_plain(x, timesteps, x_U, self.training, drop_mask, drop_mask_U, index=index, posterior=posterior)
