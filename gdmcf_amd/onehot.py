"""One-hot backbone `DNNOneHot` (reference models/DNN.py:360-477) on the HIP path -- first slice of SURVEY 8 f1.

The denoiser has two input branches: [x_t, emb] through `in_layers` and the flattened one-hot rows [x_U (two columns per
item), emb] through `in_layers2`; their hidden activations are concatenated in front of `out_layers`.  It is driven by
`GaussianDiffusionDiscrete(CatOneHot=True)` (gaussian_diffusion.py), whose discrete transition noise on the one-hot rows
is `gdmcf_onehot_noise_f32`.  Everything else is the same C-ABI kernels as the plain DNN: the input builder (once per
branch), the dense layers, the fused loss epilogue, the weight / input gradient GEMMs and the embedding-branch backward
(once per branch, the two `emb_layer` gradients are added).  The concatenation costs nothing: the last layer of each
branch writes straight into its column range of one [B, h1 + h2] buffer, and the gradient of that buffer is read back
by column range.

`gemm_dtype="bf16"` rounds the GEMM operands to bf16 on chip (from the f32 tensors: no bf16 shadows here).  Data parallel: the engine hands every gradient to `DataParallelStep`'s sink as soon as its
kernels are enqueued (overlapped all-reduce, or the sharded optimiser with its all-gathers waited for at the end of the
step).  Constructor, parameter names and initialisation draw order are the reference's, so checkpoints interchange.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .engine import DenoiserEngine


def _ceil64(n):
    return (int(n) + 63) // 64 * 64


class _Bufs:
    pass


class _OneHotTrainLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, spec, *params):
        loss = eng.train_forward(spec)
        ctx.eng, ctx.version = eng, eng.version
        return loss

    @staticmethod
    def backward(ctx, gloss):
        eng = ctx.eng
        if ctx.version != eng.version:
            raise RuntimeError("gdmcf_amd: activations were overwritten by a later forward; "
                               "call backward before the next training_losses/forward")
        return (None, None, *eng.train_backward(gloss))


class OneHotEngine:
    _eps_target = DenoiserEngine._eps_target  # (target, alpha, rowdiv) of the eps parameterisation: one launch
    supports_grad_sink = True  # parallel.DataParallelStep may install `grad_sink` (overlapped gradient exchange)
    fused_opt = None

    def __init__(self, model):
        self.model = model
        self.lib = _lib.load()
        self.E = int(model.time_emb_dim)
        self.I = int(model.in_dims[0])
        self.version = 0
        self.seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0
        self._bufs = {}
        self._saved = None
        # data parallel: called as grad_sink(param, grad) the moment a gradient's kernels are enqueued (see
        # engine.DenoiserEngine.grad_sink); the backward then returns None for that parameter
        self.grad_sink = None

    def manual_seed(self, seed):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0

    def _precision(self):
        """GDMCF_GEMM_F32 (0) or GDMCF_GEMM_BF16 (1: operands rounded to bf16 on their way to LDS -- no shadows here)."""
        return {"f32": 0, "bf16": 1, "f32x3": 2}[getattr(self.model, "gemm_dtype", "f32")]

    # -- layers -------------------------------------------------------------------------------------------------------
    def _chains(self):
        m = self.model
        br1 = [(l.weight, l.bias, 1) for l in m.in_layers]
        br2 = [(l.weight, l.bias, 1) for l in m.in_layers2]
        n_out = len(m.out_layers)
        out = [(l.weight, l.bias, 1 if i != n_out - 1 else 0) for i, l in enumerate(m.out_layers)]
        for w, b, _ in br1 + br2 + out:
            _lib.require_gpu(w, "DNNOneHot parameters")
            if not (w.is_contiguous() and b.is_contiguous() and w.dtype == torch.float32):
                raise RuntimeError("gdmcf_amd: DNNOneHot parameters must be contiguous float32")
        return br1, br2, out

    def buffers(self, B, device):
        key = (B, str(device))
        b = self._bufs.get(key)
        if b is not None:
            return b
        lib, I, E = self.lib, self.I, self.E
        br1, br2, out = self._chains()
        f32 = dict(dtype=torch.float32, device=device)
        b = _Bufs()
        b.ld1, b.ld2 = _ceil64(I + E), _ceil64(2 * I + E)
        b.xin1 = torch.zeros(B, b.ld1, **f32)
        b.xin2 = torch.zeros(B, b.ld2, **f32)
        b.xU = torch.zeros(B, 2 * I, **f32)
        b.temb = torch.zeros(B, max(E, 1), **f32)
        b.rownorm = torch.zeros(B, **f32)
        b.h1, b.h2 = br1[-1][0].shape[0], br2[-1][0].shape[0]
        b.hcat = torch.zeros(B, _ceil64(b.h1 + b.h2), **f32)
        b.dhcat = torch.zeros_like(b.hcat)
        mk = lambda chain: [torch.zeros(B, _ceil64(w.shape[0]), **f32) for (w, _, _) in chain[:-1]]
        b.acts1, b.acts2, b.acts_out = mk(br1), mk(br2), mk(out)
        b.dz1, b.dz2, b.dz_out = mk(br1), mk(br2), mk(out)
        b.hs = torch.zeros(B, _ceil64(out[-1][0].shape[1]), **f32)  # row-scaled input of the loss layer (its weight gradient)
        b.ldi = _ceil64(I)
        b.diff = torch.zeros(B, b.ldi, **f32)
        b.xt = None
        b.rowpart = torch.zeros(B, lib.gdmcf_loss_tiles(out[-1][0].shape[0]), **f32)
        b.rowsum = torch.zeros(B, **f32)
        b.gradcoef = torch.zeros(B, **f32)
        b.rowdiv_mse = torch.full((B,), float(I), **f32)
        b.lu = torch.zeros(B, dtype=torch.float64, device=device)
        n_first = max(br1[0][0].shape[0], br2[0][0].shape[0])
        b.demb = torch.zeros((B + n_first) * max(E, 1), **f32)
        ws = 0
        for chain in (br1, br2, out):
            for (w, _, _) in chain:
                ws = max(ws, lib.gdmcf_linear_ws_bytes(B, w.shape[0], w.shape[1]))
        b.ws_bytes = int(ws)
        b.ws = torch.empty(max(ws, 256), dtype=torch.uint8, device=device)
        self._bufs[key] = b
        return b

    # -- input builders ---------------------------------------------------------------------------------------------
    def _prep(self, bufs, x, I, xin, ts, ca, cb, noise, drop_mask, training, xt_out=None):
        """gdmcf_dnn_prep_input_f32 on a [B, I] operand (the rows themselves, or their [B, 2I] one-hot image)."""
        m, lib = self.model, self.lib
        B = x.shape[0]
        noise_mode = 0
        if ca is not None:
            noise_mode = 1 if noise is not None else 2
            if noise is not None and (noise.dtype != torch.float32 or noise.stride(-1) != 1):
                noise = noise.float().contiguous()
        p = float(m.drop.p)
        drop_mode, keep = 0, None
        if drop_mask is not None:
            drop_mode = 1
            keep = drop_mask.reshape(B, -1)
            keep = (keep if keep.dtype == torch.uint8 else (keep != 0).to(torch.uint8)).contiguous()
        elif training and p > 0.0:
            drop_mode = 2
        self.offset += 1
        _lib.check(lib.gdmcf_dnn_prep_input_f32(
            x.data_ptr(), x.stride(0), _lib.ptr(ts), _lib.ptr(ca), _lib.ptr(cb), noise_mode, _lib.ptr(noise),
            noise.stride(0) if noise is not None else 0, drop_mode, _lib.ptr(keep),
            keep.stride(0) if keep is not None else 0, p, self.seed, self.offset, int(bool(m.norm)),
            m.emb_layer.weight.data_ptr(), m.emb_layer.bias.data_ptr(), self.E, B, I, xin.data_ptr(), xin.stride(0),
            _lib.ptr(xt_out), xt_out.stride(0) if xt_out is not None else 0, bufs.temb.data_ptr(),
            bufs.rownorm.data_ptr(), _lib.stream_ptr()))
        return noise, keep

    def onehot_rows(self, x0, ts_U, sampled, discrete, out=None):
        """x_tU of the reference (:841-849 / :672-686) as the [B, 2I] float image the second branch reads."""
        B = x0.shape[0]
        if x0.dtype != torch.float32 or x0.stride(-1) != 1:
            x0 = x0.float().contiguous()
        if out is None:
            out = torch.empty(B, 2 * self.I, dtype=torch.float32, device=x0.device)
        s8 = None
        if sampled is not None:
            s8 = (sampled if sampled.dtype == torch.uint8 else (sampled != 0).to(torch.uint8)).contiguous()
        elif ts_U is not None:
            ts_U = ts_U.to(device=x0.device, dtype=torch.int64).contiguous()
        self.offset += 1
        _lib.check(self.lib.gdmcf_onehot_noise_f32(
            x0.data_ptr(), x0.stride(0), _lib.ptr(ts_U), B, self.I, float(discrete), _lib.ptr(s8),
            s8.stride(0) if s8 is not None else 0, self.seed, self.offset, out.data_ptr(), out.stride(0), None, 0,
            _lib.stream_ptr()))
        return out, (x0, s8, ts_U)

    def _chain_forward(self, bufs, chain, acts, A, lda, B, last_out, last_ld):
        """All layers of one branch; the last one writes to (last_out pointer, last_ld)."""
        lib, st = self.lib, _lib.stream_ptr()
        for li, (w, bias, act) in enumerate(chain):
            N, K = w.shape
            last = li == len(chain) - 1
            optr, old = (last_out, last_ld) if last else (acts[li].data_ptr(), acts[li].stride(0))
            _lib.check(lib.gdmcf_linear_fwd_f32(A.data_ptr() if torch.is_tensor(A) else A, lda, w.data_ptr(), w.stride(0),
                                                bias.data_ptr(), act, B, N, K, optr, old, bufs.ws.data_ptr(),
                                                bufs.ws_bytes, st))
            if not last:
                A, lda = acts[li], acts[li].stride(0)

    def _hidden(self, bufs, br1, br2, out, B):
        """Both branches into hcat, then all out layers but the last; returns (A, lda) feeding the last layer."""
        self._chain_forward(bufs, br1, bufs.acts1, bufs.xin1, bufs.ld1, B, bufs.hcat.data_ptr(), bufs.hcat.stride(0))
        self._chain_forward(bufs, br2, bufs.acts2, bufs.xin2, bufs.ld2, B, bufs.hcat.data_ptr() + 4 * bufs.h1,
                            bufs.hcat.stride(0))
        A, lda = bufs.hcat, bufs.hcat.stride(0)
        if len(out) > 1:
            lib, st = self.lib, _lib.stream_ptr()
            for li, (w, bias, act) in enumerate(out[:-1]):
                N, K = w.shape
                o = bufs.acts_out[li]
                _lib.check(lib.gdmcf_linear_fwd_f32(A.data_ptr(), lda, w.data_ptr(), w.stride(0), bias.data_ptr(), act, B,
                                                    N, K, o.data_ptr(), o.stride(0), bufs.ws.data_ptr(), bufs.ws_bytes, st))
                A, lda = o, o.stride(0)
        return A, lda

    # -- fused training forward / backward ----------------------------------------------------------------------------
    def train_forward(self, spec):
        prev = self.lib.gdmcf_gemm_precision(self._precision())
        try:
            return self._train_forward(spec)
        finally:
            self.lib.gdmcf_gemm_precision(prev)

    def _train_inputs(self, spec, bufs):
        """Both branch inputs (xin1: noised rows, xin2: one-hot image; dropout, normalize, embedding columns) and the
        loss target.  Returns (x0, target, alpha, rowdiv, keepalive)."""
        x0, ts = spec["x_start"], spec["ts"]
        B, dev = x0.shape[0], x0.device
        if x0.dtype != torch.float32 or x0.stride(-1) != 1:
            x0 = x0.float().contiguous()
        eps_mode = spec["eps_mode"]
        xt_out = None
        if eps_mode:
            if bufs.xt is None:
                bufs.xt = torch.zeros(B, bufs.ldi, dtype=torch.float32, device=dev)
            xt_out = bufs.xt
        training = self.model.training
        _, s8 = self.onehot_rows(x0, spec["ts_U"], spec["sampled"], spec["discrete"], out=bufs.xU)
        noise, keep1 = self._prep(bufs, x0, self.I, bufs.xin1, ts, spec["ca"], spec["cb"], spec["noise"], spec["drop_mask"],
                                  training, xt_out=xt_out)
        _, keep2 = self._prep(bufs, bufs.xU, 2 * self.I, bufs.xin2, ts, None, None, None, spec["drop_mask_U"], training)
        alpha = None
        if eps_mode:
            target, alpha, rowdiv = self._eps_target(bufs, spec, ts, x0, noise)
        else:
            target, rowdiv = x0, bufs.rowdiv_mse
        return x0, target, alpha, rowdiv, (s8, noise, keep1, keep2)

    def _loss_layer(self, spec, bufs, B, A_ptr, lda, W_ptr, ldw, bias_ptr, N, K, target, alpha, rowdiv):
        """Last product fused with the per-row loss, then the float64 loss tail (weights, history FIFO, 1/pt)."""
        lib, st = self.lib, _lib.stream_ptr()
        ts, pt = spec["ts"], spec["pt"]
        _lib.check(lib.gdmcf_linear_loss_fwd_f32(A_ptr, lda, W_ptr, ldw, bias_ptr, target.data_ptr(), target.stride(0),
                                                 _lib.ptr(alpha), B, N, K, None, 0, bufs.diff.data_ptr(), bufs.ldi,
                                                 bufs.rowpart.data_ptr(), bufs.rowsum.data_ptr(), st))
        loss = torch.empty(B, dtype=torch.float64, device=ts.device)
        _lib.check(lib.gdmcf_row_loss_finish_f64(bufs.rowsum.data_ptr(), rowdiv.data_ptr(), _lib.ptr(alpha), ts.data_ptr(),
                                                 spec["weight_t"].data_ptr(), pt.data_ptr(), B, spec["T"], spec["H"],
                                                 spec["Lt_history"].data_ptr(), spec["Lt_count"].data_ptr(),
                                                 int(spec["update_history"]), bufs.lu.data_ptr(), loss.data_ptr(),
                                                 bufs.gradcoef.data_ptr(), st))
        return loss

    def _train_forward(self, spec):
        B, dev = spec["x_start"].shape[0], spec["x_start"].device
        br1, br2, out = self._chains()
        bufs = self.buffers(B, dev)
        self.version += 1
        x0, target, alpha, rowdiv, keep = self._train_inputs(spec, bufs)
        A, lda = self._hidden(bufs, br1, br2, out, B)
        w, bias, _ = out[-1]
        loss = self._loss_layer(spec, bufs, B, A.data_ptr(), lda, w.data_ptr(), w.stride(0), bias.data_ptr(), w.shape[0],
                                w.shape[1], target, alpha, rowdiv)
        self._saved = dict(B=B, bufs=bufs, chains=(br1, br2, out), keepalive=(x0, keep, target, alpha, rowdiv, spec["pt"]))
        return loss

    def train_backward(self, gloss):
        prev = self.lib.gdmcf_gemm_precision(self._precision())
        try:
            return self._train_backward(gloss)
        finally:
            self.lib.gdmcf_gemm_precision(prev)

    # -- backward building blocks -------------------------------------------------------------------------------------
    def _rowscale_of(self, bufs, gloss):
        if isinstance(gloss, float):  # mean reduction: the same upstream gradient 1/B on every row
            return bufs.gradcoef * gloss
        return (gloss.to(torch.float32) * bufs.gradcoef).contiguous()

    def _weight_grad(self, bufs, B, w, bias, dz_ptr, lddz, rs, A_ptr, lda):
        """(dW, db) of one layer -- or (None, None) once handed to the data-parallel gradient sink."""
        lib, st = self.lib, _lib.stream_ptr()
        N, K = w.shape
        dW = torch.empty_like(w)
        db = torch.empty_like(bias) if bias is not None else None
        scol = 0
        if rs is not None:  # (rs . dZ)^T A == dZ^T (rs . A): scale the small activation instead of the big dZ
            _lib.check(lib.gdmcf_rowscale_f32(A_ptr, lda, rs.data_ptr(), B, K, bufs.hs.data_ptr(), bufs.hs.stride(0), st))
            A_ptr, lda = bufs.hs.data_ptr(), bufs.hs.stride(0)
            scol = int(lda > K)  # the copy's column K then holds the row scale: db comes out of the product
        _lib.check(lib.gdmcf_linear_bwd_weight_f32(dz_ptr, lddz, A_ptr, lda, _lib.ptr(rs), scol, B, N, K, dW.data_ptr(),
                                                   dW.stride(0), _lib.ptr(db), 0, st))
        if self.grad_sink is not None and bias is not None:
            self.grad_sink(w, dW)
            self.grad_sink(bias, db)
            return None, None
        return dW, db

    def _input_grad(self, bufs, B, W_ptr, ldw, N, K, dz_ptr, lddz, rs, A_ptr, lda, act_prev, d_ptr, ldd):
        _lib.check(self.lib.gdmcf_linear_bwd_input_f32(dz_ptr, lddz, W_ptr, ldw, _lib.ptr(rs), A_ptr, lda, act_prev, B, N, K,
                                                       d_ptr, ldd, bufs.ws.data_ptr(), bufs.ws_bytes, _lib.stream_ptr()))

    def _branch_backward(self, bufs, B, chain, acts, dzs, xin, ldx, I_cols, dz_ptr, lddz):
        """One input branch from d(pre-activation of its last layer): hidden layers, then the timestep-embedding columns
        of its first layer.  Returns ([(dW, db)...], dWe, dbe)."""
        m = self.model
        grads = [None] * len(chain)
        dWe = dbe = None
        for li in range(len(chain) - 1, -1, -1):
            w, bias, _ = chain[li]
            N, K = w.shape
            if li > 0:
                A_prev = acts[li - 1]
                grads[li] = self._weight_grad(bufs, B, w, bias, dz_ptr, lddz, None, A_prev.data_ptr(), A_prev.stride(0))
                self._input_grad(bufs, B, w.data_ptr(), w.stride(0), N, K, dz_ptr, lddz, None, A_prev.data_ptr(),
                                 A_prev.stride(0), chain[li - 1][2], dzs[li - 1].data_ptr(), dzs[li - 1].stride(0))
                dz_ptr, lddz = dzs[li - 1].data_ptr(), dzs[li - 1].stride(0)
            else:
                grads[li] = self._weight_grad(bufs, B, w, bias, dz_ptr, lddz, None, xin.data_ptr(), ldx)
                dWe, dbe = torch.empty_like(m.emb_layer.weight), torch.empty_like(m.emb_layer.bias)
                _lib.check(self.lib.gdmcf_emb_bwd_f32(dz_ptr, lddz, w.data_ptr(), w.stride(0), I_cols, self.E,
                                                      bufs.temb.data_ptr(), B, N, bufs.demb.data_ptr(), dWe.data_ptr(),
                                                      dbe.data_ptr(), _lib.stream_ptr()))
        return grads, dWe, dbe

    def _branches_backward(self, bufs, B, br1, br2):
        """Both branches from bufs.dhcat (gradient of the pre-activations behind hcat[:, :h1+h2]); emb_layer's two
        gradients are added.  Returns the list [dWe, dbe, in_layers..., in_layers2...]."""
        m = self.model
        lddz = bufs.dhcat.stride(0)
        g1, dWe1, dbe1 = self._branch_backward(bufs, B, br1, bufs.acts1, bufs.dz1, bufs.xin1, bufs.ld1, self.I,
                                               bufs.dhcat.data_ptr(), lddz)
        g2, dWe2, dbe2 = self._branch_backward(bufs, B, br2, bufs.acts2, bufs.dz2, bufs.xin2, bufs.ld2, 2 * self.I,
                                               bufs.dhcat.data_ptr() + 4 * bufs.h1, lddz)
        dWe, dbe = dWe1 + dWe2, dbe1 + dbe2
        if self.grad_sink is not None:
            self.grad_sink(m.emb_layer.weight, dWe)
            self.grad_sink(m.emb_layer.bias, dbe)
            dWe = dbe = None
        res = [dWe, dbe]
        for g in g1 + g2:
            res += [g[0], g[1]]
        return res

    def _train_backward(self, gloss):
        """Gradients in model.parameters() order: emb_layer (w, b), in_layers..., in_layers2..., out_layers..."""
        sv = self._saved
        if sv is None:
            raise RuntimeError("gdmcf_amd: train_backward without a preceding training_losses")
        bufs, B = sv["bufs"], sv["B"]
        br1, br2, out = sv["chains"]
        # ---- out layers: from the loss layer down to the concatenated hidden activation
        g_out = [None] * len(out)
        dz_ptr, lddz, rs = bufs.diff.data_ptr(), bufs.ldi, self._rowscale_of(bufs, gloss)
        for li in range(len(out) - 1, -1, -1):
            w, bias, _ = out[li]
            if li > 0:
                A_prev, act_prev, dprev = bufs.acts_out[li - 1], out[li - 1][2], bufs.dz_out[li - 1]
            else:
                A_prev, act_prev, dprev = bufs.hcat, 1, bufs.dhcat  # both branches end in tanh
            g_out[li] = self._weight_grad(bufs, B, w, bias, dz_ptr, lddz, rs, A_prev.data_ptr(), A_prev.stride(0))
            self._input_grad(bufs, B, w.data_ptr(), w.stride(0), w.shape[0], w.shape[1], dz_ptr, lddz, rs, A_prev.data_ptr(),
                             A_prev.stride(0), act_prev, dprev.data_ptr(), dprev.stride(0))
            dz_ptr, lddz, rs = dprev.data_ptr(), dprev.stride(0), None
        res = self._branches_backward(bufs, B, br1, br2)
        for g in g_out:
            res += [g[0], g[1]]
        return res

    # -- plain forward (evaluation / reverse loop) ----------------------------------------------------------------------
    def _last_layer(self, A_ptr, lda, w_ptr, ldw, bias_ptr, act, B, N, K, x_t, posterior, bufs, st):
        """The layer that produces the model output: plain (`out`), or -- reverse loop -- with the posterior mean of
        reference gaussian_diffusion.py:451-471 / :495-498 fused into the GEMM epilogue (`posterior` = dict of per-row
        coefficient vectors c1, c2[, r1, r2][, sigma, z], want_pred): returns (x_{t-1}, pred_xstart or None)."""
        lib, dev = self.lib, x_t.device
        if posterior is None:
            res = torch.empty(B, N, dtype=torch.float32, device=dev)
            _lib.check(lib.gdmcf_linear_fwd_f32(A_ptr, lda, w_ptr, ldw, bias_ptr, act, B, N, K, res.data_ptr(), res.stride(0),
                                                bufs.ws.data_ptr(), bufs.ws_bytes, st))
            return res
        if act != 0:
            raise RuntimeError("fused posterior: the output layer must be linear")
        po = posterior
        xn = torch.empty(B, N, dtype=torch.float32, device=dev)
        pred = torch.empty(B, N, dtype=torch.float32, device=dev) if po.get("want_pred") else None
        z = po.get("z")
        _lib.check(lib.gdmcf_linear_posterior_fwd_f32(
            A_ptr, lda, w_ptr, ldw, bias_ptr, x_t.data_ptr(), x_t.stride(0), po["c1"].data_ptr(), po["c2"].data_ptr(),
            _lib.ptr(po.get("r1")), _lib.ptr(po.get("r2")), _lib.ptr(po.get("sigma")), _lib.ptr(z),
            z.stride(0) if z is not None else 0, B, N, K, xn.data_ptr(), xn.stride(0), _lib.ptr(pred),
            pred.stride(0) if pred is not None else 0, st))
        return xn, pred

    def forward_plain(self, x, timesteps, x_U, training, drop_mask=None, drop_mask_U=None, posterior=None):
        prev = self.lib.gdmcf_gemm_precision(self._precision())
        try:
            B, dev = x.shape[0], x.device
            br1, br2, out = self._chains()
            bufs = self.buffers(B, dev)
            lib, st = self.lib, _lib.stream_ptr()
            self.version += 1
            self._saved = None
            ts = timesteps.to(device=dev, dtype=torch.int64).contiguous()
            if x.dtype != torch.float32 or x.stride(-1) != 1:
                x = x.float().contiguous()
            xu = x_U.reshape(B, -1)
            if xu.shape[1] != 2 * self.I:
                raise RuntimeError("gdmcf_amd.DNNOneHot: x_U must hold two columns per item")
            if xu.dtype != torch.float32 or xu.stride(-1) != 1:
                xu = xu.float().contiguous()
            keep = (self._prep(bufs, x, self.I, bufs.xin1, ts, None, None, None, drop_mask, training),
                    self._prep(bufs, xu, 2 * self.I, bufs.xin2, ts, None, None, None, drop_mask_U, training))
            A, lda = self._hidden(bufs, br1, br2, out, B)
            w, bias, act = out[-1]
            N, K = w.shape
            res = self._last_layer(A.data_ptr(), lda, w.data_ptr(), w.stride(0), bias.data_ptr(), act, B, N, K, x, posterior,
                                   bufs, st)
            del keep
            return res
        finally:
            self.lib.gdmcf_gemm_precision(prev)


class DNNOneHot(nn.Module):
    """Drop-in for the reference DNNOneHot (models/DNN.py:360-477).  As there, `out_dims[0]` of the CALLER's list grows
    by the width of the second branch (the reference aliases and mutates it, :384-385)."""

    def __init__(self, in_dims, out_dims, emb_size, time_type="cat", norm=False, dropout=0.5, gemm_dtype="f32"):
        super().__init__()
        if gemm_dtype not in ("f32", "bf16", "f32x3"):
            raise ValueError("Unimplemented GEMM input precision %s" % gemm_dtype)
        self.gemm_dtype = gemm_dtype  # "bf16": dense products on the bf16 matrix pipe, f32 accumulate / state (§4.4)
        self.in_dims = in_dims
        self.in_dims2 = list(in_dims)
        self.in_dims2[0] *= 2
        self.out_dims = out_dims
        assert out_dims[0] == in_dims[-1], "In and out dimensions must equal to each other."
        self.time_type = time_type
        self.time_emb_dim = emb_size
        self.norm = norm
        self.emb_layer = nn.Linear(self.time_emb_dim, self.time_emb_dim)
        if self.time_type == "cat":
            in_dims_temp = [self.in_dims[0] + self.time_emb_dim] + list(self.in_dims[1:])
            in_dims_temp2 = [self.in_dims2[0] + self.time_emb_dim] + list(self.in_dims2[1:])
        else:
            raise ValueError("Unimplemented timestep embedding type %s" % self.time_type)
        out_dims_temp = self.out_dims
        out_dims_temp[0] += self.in_dims2[-1]
        self.in_layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(in_dims_temp[:-1], in_dims_temp[1:])])
        self.in_layers2 = nn.ModuleList([nn.Linear(a, b) for a, b in zip(in_dims_temp2[:-1], in_dims_temp2[1:])])
        self.out_layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(out_dims_temp[:-1], out_dims_temp[1:])])
        self.drop = nn.Dropout(dropout)  # holds p; the masks are applied inside the HIP input kernel
        self.init_weights()
        self.lrelu = torch.nn.LeakyReLU(0.1)  # present (and unused) in the reference; kept for pickling parity
        self._engine = None

    def init_weights(self):
        for layer in list(self.in_layers) + list(self.in_layers2) + list(self.out_layers) + [self.emb_layer]:
            fan_out, fan_in = layer.weight.size()
            layer.weight.data.normal_(0.0, np.sqrt(2.0 / (fan_in + fan_out)))
            layer.bias.data.normal_(0.0, 0.001)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = OneHotEngine(self)
        return self._engine

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None
        return state

    def param_list(self):
        return list(self.parameters())

    def forward(self, x, timesteps, x_U, drop_mask=None, drop_mask_U=None, posterior=None):
        """model(x_t, t, x_tU) of the reference's evaluation path.  Training goes through
        GaussianDiffusionDiscrete.training_losses (fused forward + loss with its own backward); this plain forward
        carries no autograd graph.  `posterior` (reverse loop, see OneHotEngine._last_layer): return (x_{t-1}, pred_xstart)
        with the posterior mean fused into the output GEMM instead of the raw output."""
        _lib.require_gpu(x, "DNNOneHot input")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self.training:
            raise RuntimeError("gdmcf_amd.DNNOneHot: the plain forward is not differentiable; train through "
                               "GaussianDiffusionDiscrete.training_losses (or call under torch.no_grad())")
        return self.engine.forward_plain(x, timesteps, x_U, self.training, drop_mask, drop_mask_U, posterior=posterior)
