"""Drives the C ABI (include/gdmcf_hip.h) for one DNN denoiser: owns the HBM workspaces and the
order of kernel launches for
  * the fused training forward  (q_sample -> dropout -> layers -> row loss -> f64 tail)
  * its backward                (weight / bias / embedding gradients)
  * the plain forward/backward  (model(x, t))
  * the reverse-diffusion loop  (p_sample).
PyTorch is used for device memory and streams only; every arithmetic step is a HIP kernel.

HBM layout (all float32 row-major, leading dims padded to 64 elements = 256 B):
  xin   [B, ldk]  first-layer input  [ drop(x_t) | emb(t) | 0-pad ],  ldk = ceil64(I + E)
  act_l [B, ld_l] tanh outputs of every layer but the last
  diff  [B, ldi]  alpha*out - target (the last layer's output is never stored in training)
  slabs           split-K partial sums (forward of layer 0, input-gradient of the last layer)
"""

import functools

import torch

from . import _lib


def _ceil64(n):
    return (n + 63) // 64 * 64


class _Bufs:
    pass


_GEMM_MODES = {"f32": 0, "bf16": 1, "f32x3": 2}  # include/gdmcf_hip.h GDMCF_GEMM_F32 / _BF16 / _F32X3


def _with_precision(fn):
    """Runs an engine entry point with the library's per-thread GEMM input precision set to this engine's."""
    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        prev = self.lib.gdmcf_gemm_precision(_GEMM_MODES[self.gemm_dtype])
        try:
            return fn(self, *a, **kw)
        finally:
            self.lib.gdmcf_gemm_precision(prev)
    return wrapped


class DenoiserEngine:
    supports_grad_sink = True  # parallel.DataParallelStep may install `grad_sink` (overlapped gradient exchange)

    def __init__(self, model):
        self.model = model
        self.lib = _lib.load()
        self.E = int(model.time_emb_dim)
        self.I = int(model.in_dims[0])
        self.version = 0
        self.seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0
        self._bufs = {}
        self._wshadow = {}
        self._saved = None
        # data parallel: called as grad_sink(param, grad) the moment a gradient's kernels are enqueued, so the
        # all-reduce of the big weight gradients overlaps the rest of the backward (gdmcf_amd/parallel.py).
        # When set, the engine assigns .grad itself and hands autograd None for that parameter.
        self.grad_sink = None
        # single-GPU optimiser-in-backward (FusedAdamW.fuse_into_backward): big weights are updated inside the
        # weight-gradient GEMM's epilogue; their gradient is never materialised.
        self.fused_opt = None
        # compute a layer's input gradient before its weight gradient (needed when the weight may be updated as
        # soon as its gradient exists: fused optimiser, single-process early update in parallel.DataParallelStep)
        self.input_grad_first = False
        # id(weight) -> callable that makes the current stream wait until the weight is complete.  Set by the sharded
        # data-parallel optimiser, whose all-gather of the other ranks' updated rows may still be on the wire when the
        # next step starts: train_forward waits per layer, right before the first GEMM that reads the weight, so the
        # last layer's gather travels under the first layers' GEMMs.  Every other entry point waits for all up front.
        self.weight_waiters = {}
        # gradients in persistent buffers (one per parameter, reused every step) instead of fresh tensors: constant
        # addresses for a step captured in a hipGraph (gdmcf_amd/graph.py); a `.grad` kept across steps is overwritten
        self.static_grads = False
        self._grad_bufs = {}
        # opt-in (GDMCF_GEMM_SIDE=1), single GPU: the last layer's weight-gradient GEMM on a second stream beside the
        # input-gradient GEMM (both only read dZ; each fills the other's partial last round of workgroups).  Measured
        # 1.692 -> 1.673 ms per Yelp-shape step, 4.276 -> 4.251 ms at the Amazon-Book shape, bit-identical results.  Off by
        # default: two GEMMs sharing the chip cannot be timed one by one (HIP events / rocprof show 0.44 + 0.35 ms for the
        # pair instead of 0.26 + 0.23), and the per-kernel roofline is what bench.py reports.
        import os as _os
        self._gemm_side = _os.environ.get("GDMCF_GEMM_SIDE", "0") == "1"
        self._side2 = None
        self._side2_used = False
        self._wt = {}  # id(weight) -> (weight, version, transposed copy): the reverse loop's hidden layers (see _transposed)
        self._wt_on = _os.environ.get("GDMCF_FWD_WT", "1") == "1"

    def _grad_like(self, p):
        if not self.static_grads:
            return torch.empty_like(p)
        g = self._grad_bufs.get(id(p))
        if g is None or g.shape != p.shape or g.device != p.device:
            g = self._grad_bufs[id(p)] = torch.empty_like(p)
        return g

    def _use_weight(self, w):
        fn = self.weight_waiters.pop(id(w), None) if self.weight_waiters else None
        if fn is not None:
            fn()
            rec = self._wshadow.get(id(w))
            if rec is not None and rec[1] != w._version:
                rec[0].sync()
                rec[1] = w._version

    def flush_weight_waiters(self):
        while self.weight_waiters:
            self.weight_waiters.popitem()[1]()

    @property
    def gemm_dtype(self):
        """"f32": exact-f32 MFMA products (parity path);  "bf16": operands rounded to bf16 on chip, f32 accumulate;
        "f32x3": float32 products from six bf16 MFMAs of three-term operand splits (f32-level error, gemm_split.hip)."""
        return getattr(self.model, "gemm_dtype", "f32")

    def manual_seed(self, seed):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0

    # ------------------------------------------------------------------------------------------
    # bf16 shadows (gemm_dtype == "bf16"): bf16 copies of every GEMM operand, streamed instead of the f32 tensors.
    # Activation shadows are written by the kernels that produce the activations; weight shadows are refreshed
    # here whenever the parameter's version counter moved (FusedAdamW bumps it after its raw-pointer update).
    def _shadows_on(self, bufs, layers):
        if self.gemm_dtype != "bf16":
            if bufs.shadows is not None:  # precision switched back: drop the registrations
                for sh in bufs.shadows:
                    sh.close()
                bufs.shadows = None
                bufs.sh_xin2 = None
            for sh, _ in self._wshadow.values():
                sh.close()
            self._wshadow = {}
            return
        if bufs.shadows is None:
            I, E = self.I, self.E
            mk = lambda t, cols: _lib.Bf16Shadow(t[:, :cols], sync=False)
            sh = [mk(bufs.xin, I + E), mk(bufs.diff, I)]
            for (w, _, _), a, d in zip(layers[:-1], bufs.acts, bufs.dzs):
                sh += [mk(a, w.shape[0]), mk(d, w.shape[0])]
            sh.append(mk(bufs.hs, layers[-1][0].shape[1]))
            bufs.shadows = sh
            bufs.sh_xin = sh[0]
        live = {id(w) for w, _, _ in layers}
        for key in [k for k in self._wshadow if k not in live]:  # a parameter object was replaced: drop its registration
            self._wshadow.pop(key)[0].close()
        for w, _, _ in layers:
            rec = self._wshadow.get(id(w))
            if rec is None or rec[0].ptr != w.data_ptr():
                self._use_weight(w)  # the first cast reads the weight
                if rec is not None:
                    rec[0].close()
                self._wshadow[id(w)] = [_lib.Bf16Shadow(w.detach()), w._version]
            elif id(w) in self.weight_waiters:
                continue  # still arriving: refreshed in _use_weight
            elif rec[1] != w._version:
                rec[0].sync()
                rec[1] = w._version

    def _layers(self):
        layers = self.model.layer_list()
        for w, b, _ in layers:
            _lib.require_gpu(w, "DNN parameters")
            # (rows of a weight may be further apart than its columns: FusedAdamW.fuse_into_backward seats them on 128-byte lines)
            if not (w.stride(1) == 1 and w.stride(0) >= w.shape[1] and b.is_contiguous() and w.dtype == torch.float32):
                raise RuntimeError("gdmcf_amd: DNN parameters must be float32 with unit column stride")
        return layers

    def buffers(self, B, device):
        key = (B, str(device))
        b = self._bufs.get(key)
        if b is not None:
            return b
        lib, I, E = self.lib, self.I, self.E
        layers = self.model.layer_list()
        f32 = dict(dtype=torch.float32, device=device)
        b = _Bufs()
        b.ldk = _ceil64(I + E)
        b.xin = torch.zeros(B, b.ldk, **f32)
        b.xin2 = None
        b.temb = torch.zeros(B, max(E, 1), **f32)
        b.rownorm = torch.zeros(B, **f32)
        b.acts = [torch.zeros(B, _ceil64(w.shape[0]), **f32) for (w, _, _) in layers[:-1]]
        b.dzs = [torch.zeros(B, _ceil64(w.shape[0]), **f32) for (w, _, _) in layers[:-1]]
        b.hs = torch.zeros(B, _ceil64(layers[-1][0].shape[1]), **f32)
        b.ldi = _ceil64(I)
        b.diff = torch.zeros(B, b.ldi, **f32)
        b.xt = None
        b.shadows = None  # bf16 shadows of the GEMM operands among these buffers (created on first bf16 use)
        b.rowpart = torch.zeros(B, lib.gdmcf_loss_tiles(layers[-1][0].shape[0]), **f32)
        b.rowsum = torch.zeros(B, **f32)
        b.gradcoef = torch.zeros(B, **f32)
        b.rowdiv_mse = torch.full((B,), float(I), **f32)
        b.lu = torch.zeros(B, dtype=torch.float64, device=device)
        b.demb = torch.zeros((B + layers[0][0].shape[0]) * max(E, 1), **f32)  # demb [B,E] + gathered W1[:, I:] [n0,E]
        ws = 0
        for (w, _, _) in layers:
            ws = max(ws, lib.gdmcf_linear_ws_bytes(B, w.shape[0], w.shape[1]))
        b.ws_bytes = int(ws)
        b.ws = torch.empty(max(ws, 256), dtype=torch.uint8, device=device)
        self._bufs[key] = b
        return b

    # ------------------------------------------------------------------------------------------
    def _prep(self, bufs, x, ts, ca, cb, noise, drop_mask, training, xt_out=None, xin=None):
        m, lib = self.model, self.lib
        B = x.shape[0]
        if x.dtype != torch.float32 or x.stride(-1) != 1:
            x = x.float().contiguous()
        xin = bufs.xin if xin is None else xin
        noise_mode = 0
        if ca is not None:
            noise_mode = 1 if noise is not None else 2
            if noise is not None and (noise.dtype != torch.float32 or noise.stride(-1) != 1):
                noise = noise.float().contiguous()
        p = float(m.drop.p)
        drop_mode = 0
        keep = None
        if drop_mask is not None:
            drop_mode = 1
            keep = drop_mask if drop_mask.dtype == torch.uint8 else (drop_mask != 0).to(torch.uint8)
            keep = keep.contiguous()
        elif training and p > 0.0:
            drop_mode = 2
        self.offset += 1
        rc = lib.gdmcf_dnn_prep_input_f32(
            x.data_ptr(), x.stride(0), _lib.ptr(ts), _lib.ptr(ca), _lib.ptr(cb), noise_mode, _lib.ptr(noise),
            noise.stride(0) if noise is not None else 0, drop_mode, _lib.ptr(keep),
            keep.stride(0) if keep is not None else 0, p, self.seed, self.offset, int(bool(m.norm)),
            m.emb_layer.weight.data_ptr(), m.emb_layer.bias.data_ptr(), self.E, B, self.I, xin.data_ptr(),
            xin.stride(0), _lib.ptr(xt_out), xt_out.stride(0) if xt_out is not None else 0, bufs.temb.data_ptr(),
            bufs.rownorm.data_ptr(), _lib.stream_ptr())
        _lib.check(rc)
        if xin is bufs.xin:
            bufs.xin_ones = True  # (the builder leaves 1 in column I + E: the bias column of the first layer's weight gradient)
        return x, noise, keep  # keep the (possibly converted) inputs alive until the stream has consumed them

    def _prep_csr(self, bufs, batch, ts, ca, cb, noise, drop_mask, training):
        """First-layer input straight from the device CSR rows of `batch` (data_utils.CsrBatch): no dense x0 anywhere; the
        rows' bitmaps go to bufs.x0bits for the loss epilogue."""
        m, lib = self.model, self.lib
        B, I = batch.shape
        if getattr(bufs, "x0bits", None) is None:
            bufs.x0bits = torch.zeros(B, (I + 31) // 32, dtype=torch.int32, device=batch.device)
        noise_mode = 0
        if ca is not None:
            noise_mode = 1 if noise is not None else 2
            if noise is not None and (noise.dtype != torch.float32 or noise.stride(-1) != 1):
                noise = noise.float().contiguous()
        p = float(m.drop.p)
        drop_mode, keep = 0, None
        if drop_mask is not None:
            drop_mode = 1
            keep = (drop_mask if drop_mask.dtype == torch.uint8 else (drop_mask != 0).to(torch.uint8)).contiguous()
        elif training and p > 0.0:
            drop_mode = 2
        self.offset += 1
        c = batch.csr
        _lib.check(lib.gdmcf_dnn_prep_input_csr_f32(
            c.indptr.data_ptr(), c.indices.data_ptr(), batch.row_ids.data_ptr(), _lib.ptr(ts), _lib.ptr(ca), _lib.ptr(cb),
            noise_mode, _lib.ptr(noise), noise.stride(0) if noise is not None else 0, drop_mode, _lib.ptr(keep),
            keep.stride(0) if keep is not None else 0, p, self.seed, self.offset, m.emb_layer.weight.data_ptr(),
            m.emb_layer.bias.data_ptr(), self.E, B, I, bufs.xin.data_ptr(), bufs.xin.stride(0), bufs.temb.data_ptr(),
            bufs.x0bits.data_ptr(), bufs.x0bits.stride(0), _lib.stream_ptr()))
        bufs.xin_ones = True
        return batch, noise, keep

    def _eps_target(self, bufs, spec, ts, x0, noise):
        """(target, alpha, rowdiv) of the eps parameterisation in one launch (gdmcf_eps_target_f32; reference
        gaussian_diffusion.py:344-348).  A noise tensor this package drew itself (spec["noise_owned"]) IS the target: only its
        t == 0 rows are rewritten; a caller's tensor is left alone."""
        B, dev = ts.shape[0], ts.device
        if x0.dtype != torch.float32 or x0.stride(-1) != 1:
            x0 = x0.float().contiguous()
        target = noise if spec.get("noise_owned", False) else torch.empty(B, self.I, dtype=torch.float32, device=dev)
        alpha = torch.empty(B, dtype=torch.float32, device=dev)
        rowdiv = torch.empty(B, dtype=torch.float32, device=dev)
        _lib.check(self.lib.gdmcf_eps_target_f32(
            noise.data_ptr(), noise.stride(0), bufs.xt.data_ptr(), bufs.xt.stride(0), x0.data_ptr(), x0.stride(0), ts.data_ptr(),
            spec["r1_0"].data_ptr(), spec["r2_0"].data_ptr(), int(bool(spec.get("t0_likelihood", True))), B, self.I,
            target.data_ptr(), target.stride(0), alpha.data_ptr(), rowdiv.data_ptr(), _lib.stream_ptr()))
        return target, alpha, rowdiv

    def _transposed(self, w):
        """W^T of a large weight, [in, out] row-major on 128-byte rows, cached per weight VERSION: the reverse-diffusion loop of an
        evaluation runs many batches over frozen weights, and with the weight in this orientation the hidden layer's product runs
        on the register-streaming kernel (gdmcf_linear_fwd_wt_f32): 0.224 -> 0.205 ms per step at the Yelp shape.  The transpose
        itself (one pass over the weight) is paid once per version."""
        rec = self._wt.get(id(w))
        if rec is None or rec[0] is not w or rec[1] != w._version or rec[2].device != w.device:
            n, k = w.shape
            buf = torch.zeros(k, (n + 31) // 32 * 32, dtype=torch.float32, device=w.device)
            buf[:, :n].copy_(w.detach().t())
            rec = self._wt[id(w)] = (w, w._version, buf)
        return rec[2]

    def _hidden_forward(self, bufs, layers, B, xin=None, frozen=False):
        """All layers but the last; returns (A, lda, K) feeding the last layer.  frozen: the caller runs many forward passes over
        unchanged weights (reverse loop): large layers go through their cached transposes."""
        lib, st = self.lib, _lib.stream_ptr()
        A, lda = (bufs.xin if xin is None else xin), bufs.ldk
        for li, (w, bias, act) in enumerate(layers[:-1]):
            N, K = w.shape
            out = bufs.acts[li]
            self._use_weight(w)
            if frozen and self._wt_on and self.gemm_dtype == "f32" and K >= 4096 and w.numel() >= (1 << 20):
                wt = self._transposed(w)
                _lib.check(lib.gdmcf_linear_fwd_wt_f32(A.data_ptr(), lda, wt.data_ptr(), wt.stride(0), bias.data_ptr(), act, B, N, K,
                                                       out.data_ptr(), out.stride(0), bufs.ws.data_ptr(), bufs.ws_bytes, st))
            else:
                _lib.check(lib.gdmcf_linear_fwd_f32(A.data_ptr(), lda, w.data_ptr(), w.stride(0), bias.data_ptr(), act, B,
                                                    N, K, out.data_ptr(), out.stride(0), bufs.ws.data_ptr(),
                                                    bufs.ws_bytes, st))
            A, lda = out, out.stride(0)
        return A, lda

    # ------------------------------------------------------------------------------------------
    # fused training forward / backward
    # ------------------------------------------------------------------------------------------
    @_with_precision
    def train_forward(self, spec):
        x0, ts = spec["x_start"], spec["ts"]
        csr = spec.get("csr")
        B, dev = (csr.shape[0], csr.device) if csr is not None else (x0.shape[0], x0.device)
        layers = self._layers()
        bufs = self.buffers(B, dev)
        self._shadows_on(bufs, layers)
        lib, st = self.lib, _lib.stream_ptr()
        self.version += 1
        eps_mode = spec["eps_mode"]
        xt_out = None
        if eps_mode:
            if bufs.xt is None:
                bufs.xt = torch.zeros(B, bufs.ldi, dtype=torch.float32, device=dev)
            xt_out = bufs.xt
        if csr is not None:
            keepalive = self._prep_csr(bufs, csr, ts, spec["ca"], spec["cb"], spec["noise"], spec["drop_mask"], self.model.training)
        else:
            keepalive = self._prep(bufs, x0, ts, spec["ca"], spec["cb"], spec["noise"], spec["drop_mask"],
                                   self.model.training, xt_out=xt_out)
        x0c = keepalive[0]
        alpha = None
        if eps_mode:
            # target = eps, except rows with t == 0 whose term is the x0-likelihood
            # mean((x0 - (r1*x_t - r2*eps_hat))^2 / 2)  (reference gaussian_diffusion.py:344-348)
            target, alpha, rowdiv = self._eps_target(bufs, spec, ts, x0c, keepalive[1])
        else:
            target = x0c
            rowdiv = bufs.rowdiv_mse
        A, lda = self._hidden_forward(bufs, layers, B)
        w, bias, _ = layers[-1]
        N, K = w.shape
        self._use_weight(w)
        if csr is not None:  # the target rows are bitmaps written by the CSR-fed input builder
            _lib.check(lib.gdmcf_linear_loss_fwd_bits_f32(A.data_ptr(), lda, w.data_ptr(), w.stride(0), bias.data_ptr(),
                                                          bufs.x0bits.data_ptr(), bufs.x0bits.stride(0), None, B, N, K, None,
                                                          0, bufs.diff.data_ptr(), bufs.ldi, bufs.rowpart.data_ptr(),
                                                          bufs.rowsum.data_ptr(), st))
        else:
            _lib.check(lib.gdmcf_linear_loss_fwd_f32(A.data_ptr(), lda, w.data_ptr(), w.stride(0), bias.data_ptr(),
                                                     target.data_ptr(), target.stride(0), _lib.ptr(alpha), B, N, K, None,
                                                     0, bufs.diff.data_ptr(), bufs.ldi, bufs.rowpart.data_ptr(),
                                                     bufs.rowsum.data_ptr(), st))
        loss = torch.empty(B, dtype=torch.float64, device=dev)
        pt = spec["pt"]
        # the tail also emits mean(loss) and gradcoef/B: the reference's step takes the mean next (main.py:348), and its
        # backward scales every row by 1/B -- two launches less per step (DataParallelStep uses both)
        self.last_loss_mean = torch.empty((), dtype=torch.float64, device=dev)
        if getattr(bufs, "rowscale_mean", None) is None:
            bufs.rowscale_mean = torch.zeros(B, dtype=torch.float32, device=dev)
        _lib.check(lib.gdmcf_row_loss_finish_mean_f64(bufs.rowsum.data_ptr(), rowdiv.data_ptr(), _lib.ptr(alpha),
                                                      ts.data_ptr(), spec["weight_t"].data_ptr(), pt.data_ptr(), B,
                                                      spec["T"], spec["H"], spec["Lt_history"].data_ptr(),
                                                      spec["Lt_count"].data_ptr(), int(spec["update_history"]),
                                                      bufs.lu.data_ptr(), loss.data_ptr(), bufs.gradcoef.data_ptr(),
                                                      self.last_loss_mean.data_ptr(), bufs.rowscale_mean.data_ptr(), st))
        self._saved = dict(kind="train", B=B, bufs=bufs, layers=layers, keepalive=(keepalive, target, alpha, rowdiv, pt))
        return loss

    @_with_precision
    def train_backward(self, gloss):
        """gloss: d(total)/d(loss_b) as a tensor [B] (what autograd hands over), or a Python float when every row has
        the same upstream gradient (mean reduction: 1/B) -- then no autograd graph is needed at all."""
        sv = self._saved
        if sv is None or sv.get("kind") != "train":
            raise RuntimeError("gdmcf_amd: train_backward without a preceding training_losses")
        bufs = sv["bufs"]
        if isinstance(gloss, float) and gloss == 1.0 / sv["B"] and getattr(bufs, "rowscale_mean", None) is not None:
            rowscale = bufs.rowscale_mean  # gradcoef * (float)(1/B), written by the loss tail: same bits as the product below
        elif isinstance(gloss, float):
            rowscale = bufs.gradcoef * gloss
        else:
            rowscale = (gloss.to(torch.float32) * bufs.gradcoef).contiguous()
        return self._backward(sv, bufs.diff, bufs.ldi, rowscale)

    # ------------------------------------------------------------------------------------------
    # plain forward / backward (model(x, t))
    # ------------------------------------------------------------------------------------------
    @_with_precision
    def forward_plain(self, x, timesteps, training, drop_mask=None):
        self.flush_weight_waiters()
        B, dev = x.shape[0], x.device
        layers = self._layers()
        bufs = self.buffers(B, dev)
        self._shadows_on(bufs, layers)
        lib, st = self.lib, _lib.stream_ptr()
        self.version += 1
        ts = timesteps.to(device=dev, dtype=torch.int64).contiguous()
        keepalive = self._prep(bufs, x, ts, None, None, None, drop_mask, training)
        A, lda = self._hidden_forward(bufs, layers, B)
        w, bias, act = layers[-1]
        N, K = w.shape
        out = torch.empty(B, N, dtype=torch.float32, device=dev)
        _lib.check(lib.gdmcf_linear_fwd_f32(A.data_ptr(), lda, w.data_ptr(), w.stride(0), bias.data_ptr(), act, B, N,
                                            K, out.data_ptr(), out.stride(0), bufs.ws.data_ptr(), bufs.ws_bytes, st))
        self._saved = dict(kind="plain", B=B, bufs=bufs, layers=layers, keepalive=(keepalive, ts))
        return out

    @_with_precision
    def backward_plain(self, gout):
        sv = self._saved
        g = gout.to(torch.float32)
        if g.stride(-1) != 1:
            g = g.contiguous()
        return self._backward(sv, g, g.stride(0), None)

    # ------------------------------------------------------------------------------------------
    def _backward(self, sv, dz_last, ld_last, rowscale):
        """Gradients in model.parameters() order: emb_layer (w, b), in_layers..., out_layers...
        dz_last is d(loss)/d(last layer output) up to the per-row factor `rowscale`.

        Order inside a layer: data parallel wants the weight gradient first (its all-reduce then overlaps the
        input-gradient GEMM); the fused optimiser needs the input gradient first (it reads W, which the
        weight-gradient epilogue then overwrites)."""
        lib, st = self.lib, _lib.stream_ptr()
        bufs, layers, B = sv["bufs"], sv["layers"], sv["B"]
        m = self.model
        L = len(layers)
        grads_w = [None] * L
        grads_b = [None] * L
        dWe = dbe = None
        fused = self.fused_opt if self.grad_sink is None else None
        dz, lddz, rs = dz_last, ld_last, rowscale

        def input_grad(li, w, A_prev, lda_prev, N, K):
            nonlocal dWe, dbe
            if li > 0:
                dprev = bufs.dzs[li - 1]
                act_prev = layers[li - 1][2]
                _lib.check(lib.gdmcf_linear_bwd_input_f32(dz.data_ptr(), lddz, w.data_ptr(), w.stride(0), _lib.ptr(rs),
                                                          A_prev.data_ptr(), lda_prev, act_prev, B, N, K,
                                                          dprev.data_ptr(), dprev.stride(0), bufs.ws.data_ptr(),
                                                          bufs.ws_bytes, st))
                return dprev, dprev.stride(0), None
            dWe = self._grad_like(m.emb_layer.weight)
            dbe = self._grad_like(m.emb_layer.bias)
            _lib.check(lib.gdmcf_emb_bwd_f32(dz.data_ptr(), lddz, w.data_ptr(), w.stride(0), self.I, self.E,
                                             bufs.temb.data_ptr(), B, w.shape[0], bufs.demb.data_ptr(), dWe.data_ptr(),
                                             dbe.data_ptr(), st))
            return None, 0, None

        def weight_grad(li, w, bias, A_prev, lda_prev, N, K):
            db = self._grad_like(bias)
            A_use, lda_use = A_prev, lda_prev
            # 1: column K of A_use holds the row scale (written by gdmcf_rowscale_f32 below: the copy has room for it) -- or, for
            # the first layer without a row scale, the 1 the input builder leaves in xin's first padding column
            scol = int(rs is None and li == 0 and getattr(bufs, "xin_ones", False) and lda_use > K)
            if rs is not None:
                # (rs . dZ)^T A == dZ^T (rs . A): scale the small activation instead of the big dZ
                _lib.check(lib.gdmcf_rowscale_f32(A_prev.data_ptr(), lda_prev, rs.data_ptr(), B, K, bufs.hs.data_ptr(),
                                                  bufs.hs.stride(0), st))
                A_use, lda_use = bufs.hs, bufs.hs.stride(0)
                scol = int(lda_use > K)
            fs = fused.fused_state(w) if fused is not None else None
            if fs is not None:
                if fs["exp_avg"].stride() != w.stride() or fs["exp_avg_sq"].stride() != w.stride():
                    raise RuntimeError("gdmcf_amd: the moments of a fused weight must share its leading dimension")
                _lib.check(lib.gdmcf_linear_bwd_weight_adamw_f32(
                    dz.data_ptr(), lddz, A_use.data_ptr(), lda_use, _lib.ptr(rs), scol, B, N, K, w.data_ptr(), w.stride(0),
                    fs["exp_avg"].data_ptr(), fs["exp_avg_sq"].data_ptr(), db.data_ptr(), fs["lr"], fs["beta1"],
                    fs["beta2"], fs["eps"], fs["weight_decay"], fs["step"], fs["grad_scale"], st))
                if not (self.gemm_dtype == "bf16" and _lib.shadow_info(w.data_ptr()) is not None):
                    torch.autograd.graph.increment_version(w)  # updated in the GEMM epilogue (bf16: shadow too)
                dW = None
            else:
                dW = self._grad_like(w)
                if self._gemm_side and self.grad_sink is None and li == L - 1 and L > 1:
                    # the last layer's weight-gradient GEMM on a second stream, beside the input-gradient GEMM
                    if self._side2 is None:
                        self._side2 = torch.cuda.Stream()
                    self._side2.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(self._side2):
                        _lib.check(lib.gdmcf_linear_bwd_weight_f32(dz.data_ptr(), lddz, A_use.data_ptr(), lda_use, _lib.ptr(rs),
                                                                   scol, B, N, K, dW.data_ptr(), dW.stride(0), db.data_ptr(), 0,
                                                                   _lib.stream_ptr()))
                    self._side2_used = True
                else:
                    _lib.check(lib.gdmcf_linear_bwd_weight_f32(dz.data_ptr(), lddz, A_use.data_ptr(), lda_use, _lib.ptr(rs),
                                                               scol, B, N, K, dW.data_ptr(), dW.stride(0), db.data_ptr(), 0, st))
            grads_w[li], grads_b[li] = dW, db
            if self.grad_sink is not None:
                self.grad_sink(w, dW)
                self.grad_sink(bias, db)
                grads_w[li] = grads_b[li] = None

        for li in range(L - 1, -1, -1):
            w, bias, _ = layers[li]
            N, K = w.shape
            if li > 0:
                A_prev, lda_prev = bufs.acts[li - 1], bufs.acts[li - 1].stride(0)
            else:
                A_prev, lda_prev = bufs.xin, bufs.ldk
            if fused is not None or self.input_grad_first:
                nxt = input_grad(li, w, A_prev, lda_prev, N, K)
                weight_grad(li, w, bias, A_prev, lda_prev, N, K)
            else:
                weight_grad(li, w, bias, A_prev, lda_prev, N, K)
                nxt = input_grad(li, w, A_prev, lda_prev, N, K)
            dz, lddz, rs = nxt
        if self._side2_used:  # everything that consumes the gradients is ordered behind the side stream
            torch.cuda.current_stream().wait_stream(self._side2)
            self._side2_used = False
        if self.grad_sink is not None:
            self.grad_sink(m.emb_layer.weight, dWe)
            self.grad_sink(m.emb_layer.bias, dbe)
            dWe = dbe = None
        out = [dWe, dbe]
        for li in range(L):
            out += [grads_w[li], grads_b[li]]
        return out

    # ------------------------------------------------------------------------------------------
    # reverse diffusion loop (reference gaussian_diffusion.py:161-220)
    # ------------------------------------------------------------------------------------------
    @_with_precision
    def p_sample_loop(self, x_start, steps, T, tabs32, eps_mode, sampling_noise, noise0=None, step_noise=None,
                      capture=None, draw_noise=None):
        """tabs32: dict of float32 device tables [T] (sqrt_ab, sqrt_1mab, c1, c2, r1, r2, sigma).  draw_noise(like) -> [B, I]
        float32 N(0,1): the reverse loop's th.randn_like(x_t) (reference :210-217); default: gdmcf_randn_f32 on the engine's
        Philox seed (stream 7, one offset per draw)."""
        m, lib = self.model, self.lib
        if draw_noise is None:
            def draw_noise(like):
                self.offset += 1
                return _lib.philox_randn(like.shape, like.device, self.seed, self.offset, 7)
        self.flush_weight_waiters()
        B, dev, I = x_start.shape[0], x_start.device, self.I
        layers = self._layers()
        bufs = self.buffers(B, dev)
        self._shadows_on(bufs, layers)
        st = _lib.stream_ptr()
        self.version += 1
        if bufs.xin2 is None:
            bufs.xin2 = torch.zeros_like(bufs.xin)
        if self.gemm_dtype == "bf16" and getattr(bufs, "sh_xin2", None) is None:
            bufs.sh_xin2 = _lib.Bf16Shadow(bufs.xin2[:, : I + self.E], sync=False)
            bufs.shadows.append(bufs.sh_xin2)
        norm = bool(m.norm)
        keep = []
        t_vec = torch.full((B,), max(steps - 1, 0), dtype=torch.int64, device=dev)
        ca = cb = None
        if steps > 0:
            ca, cb = tabs32["sqrt_ab"], tabs32["sqrt_1mab"]
        w, bias, _ = layers[-1]
        N, K = w.shape
        out = None

        # per-step coefficient vectors [B] and timestep vectors: expanded ONCE per (schedule, batch size) into [T, B]
        # tables whose rows are used as they stand (was 3-6 tiny launches per reverse step)
        key = (id(tabs32), T, B)
        if getattr(bufs, "step_tabs_key", None) != key:
            bufs.step_tabs = {k: tabs32[k][:T, None].expand(T, B).contiguous() for k in ("c1", "c2", "r1", "r2", "sigma")
                              if k in tabs32}
            bufs.step_ts = torch.arange(T, dtype=torch.int64, device=dev)[:, None].expand(T, B).contiguous()
            bufs.step_tabs_key = key
        stabs, step_ts = bufs.step_tabs, bufs.step_ts

        def posterior(i, n, A, lda, xt, xn):
            c1, c2 = stabs["c1"][i], stabs["c2"][i]
            r1 = r2 = sg = z = None
            if eps_mode:
                r1, r2 = stabs["r1"][i], stabs["r2"][i]
            if sampling_noise and i != 0:
                sg = stabs["sigma"][i]
                z = step_noise[n] if step_noise is not None else draw_noise(xt[:, :I])
                z = z.contiguous()
            pred = torch.empty(B, I, dtype=torch.float32, device=dev) if capture is not None else None
            _lib.check(lib.gdmcf_linear_posterior_fwd_f32(
                A.data_ptr(), lda, w.data_ptr(), w.stride(0), bias.data_ptr(), xt.data_ptr(), xt.stride(0),
                c1.data_ptr(), c2.data_ptr(), _lib.ptr(r1), _lib.ptr(r2), _lib.ptr(sg), _lib.ptr(z),
                z.stride(0) if z is not None else 0, B, N, K, xn.data_ptr(), xn.stride(0), _lib.ptr(pred),
                pred.stride(0) if pred is not None else 0, st))
            keep.append((c1, c2, r1, r2, sg, z))
            if capture is not None:
                capture.setdefault("pred_xstart", []).append(pred)
                capture.setdefault("mean", []).append(xn[:, :I].clone())

        if not norm:
            # x_t lives in the first-layer input buffers themselves: the posterior epilogue of step i writes x_{t-1}
            # straight into the other buffer's first I columns, a tiny kernel adds that step's embedding columns
            # (no per-step input builder: 2 x 55 MB less traffic per step at Yelp shape).
            cur, nxt = bufs.xin, bufs.xin2
            keep.append(self._prep(bufs, x_start, t_vec, ca, cb, noise0, None, False, xin=cur))  # x_T
            bufs.xin_ones = False  # (gdmcf_dnn_emb_cols_f32 below rewrites the padding columns with zeros)
            for n, i in enumerate(range(T - 1, -1, -1)):
                ts = step_ts[i]
                _lib.check(lib.gdmcf_dnn_emb_cols_f32(ts.data_ptr(), m.emb_layer.weight.data_ptr(),
                                                      m.emb_layer.bias.data_ptr(), self.E, B, I, cur.data_ptr(),
                                                      cur.stride(0), bufs.temb.data_ptr(), st))
                keep.append(ts)
                A, lda = self._hidden_forward(bufs, layers, B, xin=cur, frozen=True)
                out = torch.empty(B, I, dtype=torch.float32, device=dev) if i == 0 else None
                posterior(i, n, A, lda, cur, out if out is not None else nxt)
                cur, nxt = nxt, cur
        else:
            # F.normalize needs the row norms of every x_t: keep x_t separate and rebuild the layer input per step
            if bufs.xt is None:
                bufs.xt = torch.zeros(B, bufs.ldi, dtype=torch.float32, device=dev)
            xt = bufs.xt
            keep.append(self._prep(bufs, x_start, t_vec, ca, cb, noise0, None, False, xt_out=xt, xin=bufs.xin2))
            for n, i in enumerate(range(T - 1, -1, -1)):
                ts = step_ts[i]
                keep.append(self._prep(bufs, xt[:, :I], ts, None, None, None, None, False, xin=bufs.xin))
                A, lda = self._hidden_forward(bufs, layers, B, xin=bufs.xin, frozen=True)
                out = torch.empty(B, I, dtype=torch.float32, device=dev) if i == 0 else None
                xn = out if out is not None else (bufs.diff if xt is bufs.xt else bufs.xt)
                posterior(i, n, A, lda, xt, xn)
                xt = xn
        self._saved = None
        del keep
        return out
