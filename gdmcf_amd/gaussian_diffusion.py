"""Gaussian diffusion over dense user-interaction rows with the reference's interface
(reference models/gaussian_diffusion.py:54-547): same constructor, `training_losses`, `p_sample`,
`q_sample`, `sample_timesteps`, `SNR`, the schedule tables as float64 attributes, the
`Lt_history` / `Lt_count` importance-sampling state.

Everything per-element runs in HIP kernels through the C ABI (include/gdmcf_hip.h); this file is
the host-side mirror of the reference class.  Extra keyword-only arguments (`ts`, `pt`, `noise`,
`drop_mask`, ...) inject the randomness explicitly -- that is how the parity tests compare with
the CPU oracle on identical inputs.  Without them noise and dropout come from an in-kernel
Philox stream and are never materialised in HBM.
"""
import enum

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .DNN import DNN

_SCHEDULE_KIND = {"linear": 0, "linear-var": 1, "cosine": 2, "binomial": 3}


class ModelMeanType(enum.Enum):
    START_X = enum.auto()  # the model predicts x_0
    EPSILON = enum.auto()  # the model predicts epsilon


class _TrainLoss(torch.autograd.Function):
    """loss[B] (float64) of the fused q_sample -> denoiser -> weighted row-MSE path."""

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


class GaussianDiffusion(nn.Module):
    def __init__(self, mean_type, noise_schedule, noise_scale, noise_min, noise_max, steps, device,
                 history_num_per_term=10, beta_fixed=True, discrete=0.99, CatOneHot=False, epps=None, args=None):
        super().__init__()
        if CatOneHot and not getattr(self, "_onehot_ok", False):
            raise NotImplementedError("CatOneHot is built for GaussianDiffusionDiscrete only (SURVEY 8 f1); the base "
                                      "class's own one-hot branch (:294-302) is not")
        self.mean_type = mean_type
        self.noise_schedule = noise_schedule
        self.noise_scale = noise_scale
        self.noise_min = noise_min
        self.noise_max = noise_max
        self.steps = steps
        self.device = torch.device(device)
        self.discrete = discrete
        self.CatOneHot = CatOneHot
        self.gcn = 0
        self.indexIn = None
        self.history_num_per_term = history_num_per_term
        self.Lt_history = torch.zeros(steps, history_num_per_term, dtype=torch.float64, device=self.device)
        self.Lt_count = torch.zeros(steps, dtype=torch.int64, device=self.device)
        self.update_history = True  # data-parallel wrappers switch this off and replay the gathered batch
        self.rng = "philox"  # "philox": in-kernel noise/dropout;  "torch": torch.randn / torch.bernoulli
        if noise_scale != 0.0:
            if noise_schedule not in _SCHEDULE_KIND:
                raise NotImplementedError(f"unknown beta schedule: {noise_schedule}!")
            self.beta_fixed = beta_fixed
            self.calculate_for_diffusion()
        else:  # no noising: the reference builds no tables at all (:86-89); only the unweighted loss is defined
            self._t32 = {}
            self._weights = {"one": torch.ones(steps, dtype=torch.float64, device=self.device)}

    # -- tables -----------------------------------------------------------------------------
    def calculate_for_diffusion(self):
        """All float64 schedule tables of the reference (:132-159: alphas_cumprod(_prev/_next), their roots and logs,
        posterior variance / clipped log variance / mean coefficients), built by gdmcf_schedule_build."""
        tabs = _lib.schedule_tables(_SCHEDULE_KIND[self.noise_schedule], self.noise_scale, self.noise_min, self.noise_max,
                                    self.steps, self.beta_fixed)
        for name, row in zip(_lib.TABLE_NAMES, tabs):
            setattr(self, name, torch.from_numpy(row.copy()).to(self.device))
        self._derive_tables()

    def get_betas(self):
        kind = _SCHEDULE_KIND.get(self.noise_schedule)
        if kind is None:
            raise NotImplementedError(f"unknown beta schedule: {self.noise_schedule}!")
        return _lib.schedule_tables(kind, self.noise_scale, self.noise_min, self.noise_max, self.steps, False)[0]

    def _derive_tables(self):
        """float32 copies (the reference casts to f32 in _extract_into_tensor, :544) and the
        float64 per-timestep loss weights (:339-350)."""
        f = lambda t: t.to(torch.float32).contiguous()
        t = torch.arange(self.steps, device=self.device)
        self._t32 = dict(
            sqrt_ab=f(self.sqrt_alphas_cumprod), sqrt_1mab=f(self.sqrt_one_minus_alphas_cumprod),
            c1=f(self.posterior_mean_coef1), c2=f(self.posterior_mean_coef2),
            r1=f(self.sqrt_recip_alphas_cumprod), r2=f(self.sqrt_recipm1_alphas_cumprod),
            sigma=torch.exp(0.5 * f(self.posterior_log_variance_clipped)).contiguous())
        w_x0 = torch.where(t == 0, 1.0, self.SNR(t - 1) - self.SNR(t))
        w_eps = (1 - self.alphas_cumprod) / ((1 - self.alphas_cumprod_prev) ** 2 * (1 - self.betas))
        w_eps = torch.where(t == 0, 1.0, w_eps)
        self._weights = {"x0": w_x0.contiguous(), "eps": w_eps.contiguous(),
                         "one": torch.ones(self.steps, dtype=torch.float64, device=self.device)}

    def SNR(self, t):
        return self.alphas_cumprod[t] / (1 - self.alphas_cumprod[t])

    def _draw_noise(self, like, eps_mode=False, stream_id=4):
        """`th.randn_like(x)` of the reference (:328-331 the eps target, :210-217 / :696-703 the reverse loop's noise) as a device
        draw: gdmcf_randn_f32 (Philox4x32-10 keyed by torch.initial_seed(), one offset per call) unless rng == "torch"."""
        if self.rng == "torch":
            return torch.randn(like.shape, dtype=torch.float32, device=like.device)
        self._randn_calls = getattr(self, "_randn_calls", 0) + 1
        return _lib.philox_randn(like.shape, like.device, int(torch.initial_seed()) & (2 ** 63 - 1), self._randn_calls, stream_id)

    def _extract_into_tensor(self, arr, timesteps, broadcast_shape):
        res = arr.to(timesteps.device)[timesteps].float()
        while len(res.shape) < len(broadcast_shape):
            res = res[..., None]
        return res.expand(broadcast_shape)

    # -- timesteps ---------------------------------------------------------------------------
    def importance_probs(self, uniform_prob=0.001):
        Lt_sqrt = torch.sqrt(torch.mean(self.Lt_history ** 2, axis=-1))
        pt_all = Lt_sqrt / torch.sum(Lt_sqrt)
        pt_all = pt_all * (1 - uniform_prob)
        pt_all = pt_all + uniform_prob / len(pt_all)
        return pt_all

    def sample_timesteps(self, batch_size, device, method="uniform", uniform_prob=0.001):
        """(t, pt) as the reference (:373-397).  'importance' runs as ONE HIP kernel that also takes the
        "history not full yet -> uniform" branch on the device (no host sync, no torch micro-kernels)."""
        if method == "importance":
            device = torch.device(device)
            if device.type != "cuda":
                raise RuntimeError("gdmcf_amd: sample_timesteps must run on the MI355X; the HIP path has no CPU fallback")
            t = torch.empty(batch_size, dtype=torch.int64, device=device)
            pt = torch.empty(batch_size, dtype=torch.float64, device=device)
            self._ts_calls = getattr(self, "_ts_calls", 0) + 1
            _lib.check(_lib.load().gdmcf_sample_timesteps(
                self.Lt_history.data_ptr(), self.Lt_count.data_ptr(), self.steps, self.history_num_per_term,
                batch_size, float(uniform_prob), int(torch.initial_seed()) & (2 ** 63 - 1), self._ts_calls,
                t.data_ptr(), pt.data_ptr(), None, _lib.stream_ptr()))
            return t, pt
        elif method == "uniform":
            t = torch.randint(0, self.steps, (batch_size,), device=device).long()
            pt = torch.ones_like(t).float()
            return t, pt
        raise ValueError

    # -- q_sample ----------------------------------------------------------------------------
    def q_sample(self, x_start, t, noise=None):
        """x_t = sqrt(abar_t) x_0 + sqrt(1-abar_t) eps  (reference :399-407), HIP kernel."""
        _lib.require_gpu(x_start, "x_start")
        if noise is not None:
            assert noise.shape == x_start.shape
        lib = _lib.load()
        B, I = x_start.shape
        x = x_start.float().contiguous()
        ldo = (I + 3) // 4 * 4
        out = torch.empty(B, ldo, dtype=torch.float32, device=x.device)
        t = t.to(device=x.device, dtype=torch.int64).contiguous()
        nz = None if noise is None else noise.float().contiguous()
        self._q_calls = getattr(self, "_q_calls", 0) + 1
        rc = lib.gdmcf_dnn_prep_input_f32(
            x.data_ptr(), x.stride(0), t.data_ptr(), self._t32["sqrt_ab"].data_ptr(),
            self._t32["sqrt_1mab"].data_ptr(), 1 if nz is not None else 2, _lib.ptr(nz),
            nz.stride(0) if nz is not None else 0, 0, None, 0, 0.0, int(torch.initial_seed()) & (2 ** 63 - 1),
            (1 << 40) + self._q_calls, 0, None, None, 0, B, I, out.data_ptr(), ldo, None, 0, None, None,
            _lib.stream_ptr())
        _lib.check(rc)
        return out[:, :I]

    # -- the reference's per-step pieces, for callers that use them directly -----------------------
    # (p_sample itself runs them fused into the last layer's epilogue: gdmcf_linear_posterior_fwd_f32)
    def _predict_xstart_from_eps(self, x_t, t, eps):
        """reference :518-523"""
        assert x_t.shape == eps.shape
        _lib.require_gpu(x_t, "x_t")
        return (self._extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - self._extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * eps)

    def q_posterior_mean_variance(self, x_start, x_t, t):
        """q(x_{t-1} | x_t, x_0): (mean, variance, clipped log variance), reference :451-471"""
        assert x_start.shape == x_t.shape
        _lib.require_gpu(x_t, "x_t")
        posterior_mean = (self._extract_into_tensor(self.posterior_mean_coef1, t, x_t.shape) * x_start
                          + self._extract_into_tensor(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        posterior_variance = self._extract_into_tensor(self.posterior_variance, t, x_t.shape)
        posterior_log_variance_clipped = self._extract_into_tensor(self.posterior_log_variance_clipped, t, x_t.shape)
        assert (posterior_mean.shape[0] == posterior_variance.shape[0] == posterior_log_variance_clipped.shape[0]
                == x_start.shape[0])
        return posterior_mean, posterior_variance, posterior_log_variance_clipped

    def p_mean_variance(self, model, x, t):
        """One reverse step's distribution parameters, reference :473-515 (the denoiser forward is the HIP path)."""
        B, C = x.shape[:2]
        assert t.shape == (B,)
        model_output = model(x, t)
        model_variance = self._extract_into_tensor(self.posterior_variance, t, x.shape)
        model_log_variance = self._extract_into_tensor(self.posterior_log_variance_clipped, t, x.shape)
        if self.mean_type == ModelMeanType.START_X:
            pred_xstart = model_output
        elif self.mean_type == ModelMeanType.EPSILON:
            pred_xstart = self._predict_xstart_from_eps(x, t, eps=model_output)
        else:
            raise NotImplementedError(self.mean_type)
        model_mean, _, _ = self.q_posterior_mean_variance(x_start=pred_xstart, x_t=x, t=t)
        assert model_mean.shape == model_log_variance.shape == pred_xstart.shape == x.shape
        return {"mean": model_mean, "variance": model_variance, "log_variance": model_log_variance,
                "pred_xstart": pred_xstart}

    # -- training ------------------------------------------------------------------------------
    def training_losses(self, model, x_start, reweight=False, index=None, *, ts=None, pt=None, noise=None,
                        drop_mask=None):
        _lib.require_gpu(x_start, "x_start")
        if not isinstance(model, DNN):
            raise TypeError("gdmcf_amd.GaussianDiffusion.training_losses needs a gdmcf_amd.DNN denoiser")
        from .data_utils import CsrBatch
        csr_batch = None
        if isinstance(x_start, CsrBatch):
            # rows left sparse: only what the CSR-fed kernels cover, otherwise densify here
            sparse_ok = (self.mean_type == ModelMeanType.START_X and not model.norm and x_start.csr.values is None
                         and self.rng == "philox")
            if sparse_ok:
                csr_batch = x_start
            else:
                x_start = x_start.dense()
        batch_size, device = x_start.size(0), x_start.device
        assert x_start.dim() == 2 and x_start.size(1) == model.in_dims[0], "x_start must be [B, n_items]"
        if ts is None:
            ts, pt = self.sample_timesteps(batch_size, device, "importance")
        ts = ts.to(device=device, dtype=torch.int64).contiguous()
        pt = pt.to(device=device, dtype=torch.float64).contiguous()
        eps_mode = self.mean_type == ModelMeanType.EPSILON
        if self.mean_type not in (ModelMeanType.START_X, ModelMeanType.EPSILON):
            raise NotImplementedError(self.mean_type)
        ca = cb = None
        noise_owned = False
        if self.noise_scale != 0.0:
            ca, cb = self._t32["sqrt_ab"], self._t32["sqrt_1mab"]
            if noise is None and (eps_mode or self.rng == "torch"):
                noise, noise_owned = self._draw_noise(x_start, eps_mode), True  # eps is the target: must exist in HBM
        elif eps_mode:
            raise NotImplementedError("noise_scale == 0 with mean_type EPSILON")
        if drop_mask is None and self.rng == "torch" and model.training and model.drop.p > 0:
            drop_mask = torch.bernoulli(torch.full_like(x_start, 1.0 - model.drop.p, dtype=torch.float32)).to(torch.uint8)
        if reweight == True:  # noqa: E712  (the reference's own test)
            if self.noise_scale == 0.0:
                raise AttributeError("GaussianDiffusion has no schedule tables (noise_scale == 0): the SNR weights of "
                                     "reweight=True do not exist -- the reference fails the same way (:340, :525-530)")
            weight_t = self._weights["eps" if eps_mode else "x0"]
        else:
            # the reference leaves `loss` undefined here (NameError); DiffRec semantics: unit weights on the mse
            # (for the eps target that also means no x0-likelihood term on the t == 0 rows)
            weight_t = self._weights["one"]
        spec = dict(x_start=None if csr_batch is not None else x_start, csr=csr_batch, ts=ts, pt=pt, ca=ca, cb=cb, noise=noise,
                    noise_owned=noise_owned,
                    drop_mask=drop_mask, eps_mode=eps_mode,
                    weight_t=weight_t, T=self.steps, H=self.history_num_per_term, Lt_history=self.Lt_history,
                    Lt_count=self.Lt_count, update_history=self.update_history, t0_likelihood=(reweight == True))  # noqa: E712
        if eps_mode:
            spec["r1_0"] = self._t32["r1"][0]
            spec["r2_0"] = self._t32["r2"][0]
        eng = model.engine
        loss = _TrainLoss.apply(eng, spec, *model.param_list())
        self.last_ts, self.last_loss_unscaled = ts, eng.buffers(batch_size, device).lu
        return {"loss": loss}

    # -- sampling --------------------------------------------------------------------------------
    def p_sample(self, model, x_start, steps, sampling_noise=False, index=None, *, noise0=None, step_noise=None,
                 capture=None):
        assert steps <= self.steps, "Too much steps in inference."
        _lib.require_gpu(x_start, "x_start")
        if not isinstance(model, DNN):
            raise TypeError("gdmcf_amd.GaussianDiffusion.p_sample needs a gdmcf_amd.DNN denoiser")
        if self.noise_scale == 0.0:
            x_t = x_start
            with torch.no_grad():
                for i in list(range(self.steps))[::-1]:
                    t = torch.full((x_t.shape[0],), i, dtype=torch.int64, device=x_t.device)
                    x_t = model(x_t, t)
            return x_t
        with torch.no_grad():
            return model.engine.p_sample_loop(x_start, steps, self.steps, self._t32,
                                              self.mean_type == ModelMeanType.EPSILON, bool(sampling_noise),
                                              noise0=noise0, step_noise=step_noise, capture=capture,
                                              draw_noise=lambda like: self._draw_noise(like, stream_id=7))


class GaussianDiffusionDiscrete(GaussianDiffusion):
    """The class the shipped main.py actually constructs (main.py:192-193; reference :552-1135).

    `CatOneHot=False`: its `training_losses` is bit-identical to `GaussianDiffusion.training_losses`, which
    is what runs here; its `p_sample` additionally samples a degree-guided one-hot graph per step (:706-744) that only
    GCN backbones consume (`graph=` of DNNOneHotEmbeddingGCN), so with a `gdmcf_amd.DNN` model the result equals the
    continuous reverse loop of the parent class.

    `CatOneHot=True` with a `gdmcf_amd.DNNOneHot` denoiser (`indexIn` False; SURVEY 8 f1, first slice): the rows are
    handed to the model a second time as one-hot pairs under the discrete transition noise of :770-831
    (`gdmcf_onehot_noise_f32`).  As in the reference, that noise uses its OWN timestep draw (:843) -- the model is
    conditioned on the second one (:865).  `indexIn = True` (set by main.py:241) selects the embedding backbone
    `gdmcf_amd.DNNOneHotEmbedding`, which needs the users' ids (`index=`) and adds 0.1 x its NT-Xent term to every row's
    loss (:886-889, :952-953); the GCN backbones are not built."""

    _onehot_ok = True

    def __init__(self, mean_type, noise_schedule, noise_scale, noise_min, noise_max, steps, device,
                 history_num_per_term=10, beta_fixed=True, discrete=0.99, CatOneHot=False, epps=0.9995, args=None):
        super().__init__(mean_type, noise_schedule, noise_scale, noise_min, noise_max, steps, device,
                         history_num_per_term=history_num_per_term, beta_fixed=beta_fixed, discrete=discrete,
                         CatOneHot=CatOneHot, epps=epps, args=args)
        self.args = args
        self.discrete_noise = True
        self.indexIn = False
        self.user_guided = bool(getattr(args, "user_guided", False))  # the only field of `args` the reference reads (:720)
        self.last_graph = None

    # -- the reference's public pieces of the discrete noise, for callers that use them directly -------------------
    def get_Qt_bar(self, alpha_bar_t):
        """[B, 2, 2] transition matrices a*I + (1-a)*[[e,1-e],[e,1-e]], e = `discrete` (reference :597-614)."""
        dev = alpha_bar_t.device
        e = self.discrete
        u_x = torch.tensor([[e, 1 - e], [e, 1 - e]], device=dev).unsqueeze(0)
        a = alpha_bar_t.unsqueeze(1).unsqueeze(1)
        return a * torch.eye(2, device=dev).unsqueeze(0) + (1 - a) * u_x

    def apply_noise(self, ts, x_start, x_base=None):
        """One-hot [B, I, 2] (int64) of a class drawn per item from row c0 of Q_bar(ts / batch_size) (reference
        :770-831); the draw runs inside gdmcf_onehot_noise_f32 (Philox), x_start is the one-hot [B, I, 2] input."""
        _lib.require_gpu(x_start, "x_start")
        B, I = x_start.shape[0], x_start.shape[1]
        x0 = x_start[..., 1].float().contiguous()  # class index of a one-hot pair
        ts = ts.to(device=x0.device, dtype=torch.int64).contiguous()
        scratch = torch.empty(B, 2 * I, dtype=torch.float32, device=x0.device)
        sampled = torch.empty(B, I, dtype=torch.uint8, device=x0.device)
        self._noise_calls = getattr(self, "_noise_calls", 0) + 1
        _lib.check(_lib.load().gdmcf_onehot_noise_f32(
            x0.data_ptr(), x0.stride(0), ts.data_ptr(), B, I, float(self.discrete), None, 0,
            int(torch.initial_seed()) & (2 ** 63 - 1), (1 << 41) + self._noise_calls, scratch.data_ptr(), scratch.stride(0),
            sampled.data_ptr(), sampled.stride(0), _lib.stream_ptr()))
        return torch.nn.functional.one_hot(sampled.long(), num_classes=2)

    def _onehot_model(self, model):
        from .onehot import DNNOneHot
        from .onehot_embedding import DNNOneHotEmbedding
        if self.indexIn:  # main.py:239-242 sets it together with the embedding backbones
            if not isinstance(model, DNNOneHotEmbedding):
                raise NotImplementedError("indexIn is built for gdmcf_amd.DNNOneHotEmbedding / DNNOneHotEmbeddingGCN only "
                                          "(the *_conti / *_time variants are not, SURVEY 8 f1)")
        elif not isinstance(model, DNNOneHot) or isinstance(model, DNNOneHotEmbedding):
            raise TypeError("gdmcf_amd.GaussianDiffusionDiscrete(CatOneHot=True) needs a gdmcf_amd.DNNOneHot denoiser "
                            "(DNNOneHotEmbedding with indexIn = True)")
        return model

    def training_losses(self, model, x_start, reweight=False, index=None, *, ts=None, pt=None, noise=None,
                        drop_mask=None, ts_U=None, sampled=None, drop_mask_U=None):
        if not self.CatOneHot:
            return super().training_losses(model, x_start, reweight, index, ts=ts, pt=pt, noise=noise, drop_mask=drop_mask)
        from .onehot import _OneHotTrainLoss
        _lib.require_gpu(x_start, "x_start")
        model = self._onehot_model(model)
        batch_size, device = x_start.size(0), x_start.device
        assert x_start.dim() == 2 and x_start.size(1) == model.in_dims[0], "x_start must be [B, n_items]"
        if sampled is None and ts_U is None:
            ts_U, _ = self.sample_timesteps(batch_size, device, "importance")  # first draw: the one-hot rows' noise level
        if ts is None:
            ts, pt = self.sample_timesteps(batch_size, device, "importance")  # second draw: what the model sees
        ts = ts.to(device=device, dtype=torch.int64).contiguous()
        pt = pt.to(device=device, dtype=torch.float64).contiguous()
        eps_mode = self.mean_type == ModelMeanType.EPSILON
        if self.mean_type not in (ModelMeanType.START_X, ModelMeanType.EPSILON):
            raise NotImplementedError(self.mean_type)
        ca = cb = None
        noise_owned = False
        if self.noise_scale != 0.0:
            ca, cb = self._t32["sqrt_ab"], self._t32["sqrt_1mab"]
            if noise is None and (eps_mode or self.rng == "torch"):
                noise, noise_owned = self._draw_noise(x_start, eps_mode), True
        elif eps_mode:
            raise NotImplementedError("noise_scale == 0 with mean_type EPSILON")
        if reweight == True:  # noqa: E712
            if self.noise_scale == 0.0:
                raise AttributeError("GaussianDiffusionDiscrete has no schedule tables (noise_scale == 0)")
            weight_t = self._weights["eps" if eps_mode else "x0"]
        else:
            weight_t = self._weights["one"]
        spec = dict(x_start=x_start, ts=ts, pt=pt, ca=ca, cb=cb, noise=noise, noise_owned=noise_owned, drop_mask=drop_mask,
                    eps_mode=eps_mode,
                    weight_t=weight_t, T=self.steps, H=self.history_num_per_term, Lt_history=self.Lt_history,
                    Lt_count=self.Lt_count, update_history=self.update_history, t0_likelihood=(reweight == True),  # noqa: E712
                    ts_U=ts_U, sampled=sampled, drop_mask_U=drop_mask_U, discrete=self.discrete, index=index)
        if eps_mode:
            spec["r1_0"] = self._t32["r1"][0]
            spec["r2_0"] = self._t32["r2"][0]
        eng = model.engine
        loss = _OneHotTrainLoss.apply(eng, spec, *model.param_list())
        self.last_ts, self.last_loss_unscaled = ts, eng.buffers(batch_size, device).lu
        return {"loss": loss}

    def p_sample(self, model, x_start, steps, sampling_noise=False, index=None, *, noise0=None, step_noise=None,
                 capture=None, sampled0=None, graph_sampled=None, graph_pick=None):
        """Reverse loop of the one-hot variant (reference :668-768).  With `indexIn` (the embedding backbones) every reverse
        step also advances the degree-guided graph of :706-729 on the device (gdmcf_graph_guided_step_u8: classes drawn
        from the accumulated graph's transition rows, one bit per user drawn from its relative degree, AND-ed when
        args.user_guided, OR-ed into the graph) and hands it to the model as `graph=` -- a uint8 [B, I] tensor of edge
        states (the reference passes its one-hot [B, I, 2] image).  `graph_sampled` [T, B, I] / `graph_pick` [T, B] (uint8)
        inject the draws (parity runs); `self.last_graph` keeps the final graph, `capture["graph"]` every step's."""
        if not self.CatOneHot:
            return super().p_sample(model, x_start, steps, sampling_noise, index, noise0=noise0, step_noise=step_noise,
                                    capture=capture)
        assert steps <= self.steps, "Too much steps in inference."
        _lib.require_gpu(x_start, "x_start")
        model = self._onehot_model(model)
        B, dev = x_start.shape[0], x_start.device
        with torch.no_grad():
            x0 = x_start.float().contiguous()
            if steps == 0:
                # the noiseless one-hot image: every true bit survives (sampled == the rows themselves)
                x_tU, keep = model.engine.onehot_rows(x0, None, (x0 != 0).to(torch.uint8), self.discrete)
                x_t = x0
            else:
                t = torch.full((B,), steps - 1, dtype=torch.int64, device=dev)
                x_tU, keep = model.engine.onehot_rows(x0, t, sampled0, self.discrete)
                x_t = self.q_sample(x0, t, noise0) if self.noise_scale != 0.0 else x0
            graph = degp = None
            if self.indexIn and self.noise_scale != 0.0:
                graph = torch.zeros(B, x0.shape[1], dtype=torch.uint8, device=dev)
                deg = x0.sum(dim=1)
                degp = (deg / deg.max()).float().contiguous()  # :710-711 (a batch without any interaction gives nan, as there)
            eps_mode = self.mean_type == ModelMeanType.EPSILON
            if self.mean_type not in (ModelMeanType.START_X, ModelMeanType.EPSILON):
                raise NotImplementedError(self.mean_type)
            # per-step coefficient vectors [T, B] of the fused posterior epilogue (reference :451-471, :495-498), once
            tabs = {k: self._t32[k][: self.steps, None].expand(self.steps, B).contiguous() for k in ("c1", "c2", "r1", "r2", "sigma")} \
                if self.noise_scale != 0.0 else None
            # models of this package fuse the posterior into their output GEMM (`posterior=` of their forward); anything
            # else with the reference's call signature takes the element-wise path below
            import inspect
            try:
                fused_posterior = "posterior" in inspect.signature(getattr(model, "forward", model)).parameters
            except (TypeError, ValueError):
                fused_posterior = False
            for n, i in enumerate(list(range(self.steps))[::-1]):
                t = torch.full((B,), i, dtype=torch.int64, device=dev)
                kw = dict(index=index) if self.indexIn else {}
                if self.noise_scale == 0.0:
                    x_t = model(x_t, t, x_tU, **kw)
                    continue
                if graph is not None:
                    self._graph_step(graph, t, degp, None if graph_sampled is None else graph_sampled[n],
                                     None if graph_pick is None else graph_pick[n])
                    kw["graph"] = graph
                    if capture is not None:
                        capture.setdefault("graph", []).append(graph.clone())
                # posterior mean (and the sampling noise) inside the epilogue of the model's output GEMM: no [B, I] model
                # output, no element-wise passes over it
                po = dict(c1=tabs["c1"][i], c2=tabs["c2"][i], want_pred=capture is not None)
                if eps_mode:
                    po.update(r1=tabs["r1"][i], r2=tabs["r2"][i])
                if sampling_noise and i != 0:
                    z = step_noise[n] if step_noise is not None else self._draw_noise(x_t, stream_id=7)
                    po.update(sigma=tabs["sigma"][i], z=z.float().contiguous())
                x_in = x_t if (x_t.dtype == torch.float32 and x_t.is_contiguous()) else x_t.float().contiguous()
                noisy = sampling_noise and i != 0
                if fused_posterior:
                    if capture is not None and noisy:
                        # parity consumers get the real posterior mean: ask the epilogue for the mean alone and add the
                        # noise here (a capture run is a test run; the product path below adds it inside the epilogue)
                        po_mean = {k: v for k, v in po.items() if k not in ("sigma", "z")}
                        mean, pred = model(x_in, t, x_tU, posterior=po_mean, **kw)
                        x_next = torch.addcmul(mean, po["sigma"][:, None], po["z"])
                    else:
                        x_next, pred = model(x_in, t, x_tU, posterior=po, **kw)
                        mean = x_next
                else:
                    # any callable with the reference's signature model(x, t, x_tU[, index, graph]) (:745-760): the same
                    # arithmetic as the fused epilogue, as element-wise passes over the model output
                    out = model(x_in, t, x_tU, **kw).float()
                    pred = po["r1"][:, None] * x_in - po["r2"][:, None] * out if eps_mode else out
                    mean = po["c1"][:, None] * pred + po["c2"][:, None] * x_in
                    x_next = torch.addcmul(mean, po["sigma"][:, None], po["z"]) if noisy else mean
                if capture is not None:
                    capture.setdefault("pred_xstart", []).append(pred)
                    capture.setdefault("mean", []).append(mean)
                x_t = x_next
            del keep
            self.last_graph = graph
            return x_t

    def _graph_step(self, graph, t, degp, sampled=None, pick=None, sampled_out=None, pick_out=None):
        """graph |= s & (user_guided ? pick : 1) for one reverse step (reference :709-727), in place, on the device."""
        B, I = graph.shape
        self._noise_calls = getattr(self, "_noise_calls", 0) + 1
        u8 = lambda v: None if v is None else v.to(device=graph.device, dtype=torch.uint8).contiguous()
        sampled, pick = u8(sampled), u8(pick)
        _lib.check(_lib.load().gdmcf_graph_guided_step_u8(
            graph.data_ptr(), graph.stride(0), t.data_ptr(), B, I, float(self.discrete), _lib.ptr(sampled),
            sampled.stride(0) if sampled is not None else 0, _lib.ptr(pick), _lib.ptr(degp), int(self.user_guided),
            int(torch.initial_seed()) & (2 ** 63 - 1), (1 << 42) + self._noise_calls, _lib.ptr(sampled_out),
            sampled_out.stride(0) if sampled_out is not None else 0, _lib.ptr(pick_out), _lib.stream_ptr()))
        return graph


# ---- module-level helpers of the reference file (kept for API parity) -------------------------------
def betas_from_linear_variance(steps, variance, max_beta=0.999):
    alpha_bar = 1 - variance
    betas = [1 - alpha_bar[0]]
    for i in range(1, steps):
        betas.append(min(1 - alpha_bar[i] / alpha_bar[i - 1], max_beta))
    return np.array(betas)


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    betas = []
    for i in range(num_diffusion_timesteps):
        t1 = i / num_diffusion_timesteps
        t2 = (i + 1) / num_diffusion_timesteps
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return np.array(betas)


def normal_kl(mean1, logvar1, mean2, logvar2):
    """KL between two diagonal gaussians (reference :1165-1192; dead code there)."""
    tensor = next((o for o in (mean1, logvar1, mean2, logvar2) if isinstance(o, torch.Tensor)), None)
    assert tensor is not None, "at least one argument must be a Tensor"
    logvar1, logvar2 = [x if isinstance(x, torch.Tensor) else torch.tensor(x).to(tensor) for x in (logvar1, logvar2)]
    return 0.5 * (-1.0 + logvar2 - logvar1 + torch.exp(logvar1 - logvar2) + ((mean1 - mean2) ** 2) * torch.exp(-logvar2))


def mean_flat(tensor):
    return tensor.mean(dim=list(range(1, len(tensor.shape))))
