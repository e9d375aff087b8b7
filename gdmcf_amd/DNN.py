"""MLP denoiser with the reference's interface (reference models/DNN.py:11-88, :1806-1825).

Same constructor, same parameter names (`emb_layer.*`, `in_layers.N.*`, `out_layers.N.*`), same
initialisation draw order, so checkpoints interchange with the reference.  The arithmetic runs in
hand-written HIP kernels through the C ABI (include/gdmcf_hip.h): the nn.Linear modules here are
parameter containers only and are never called.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .engine import DenoiserEngine


def timestep_embedding(timesteps, dim, max_period=10000):
    """Sinusoidal timestep embedding (reference models/DNN.py:1806-1825).  Host-side helper kept
    for API parity; the hot path computes it inside gdmcf_dnn_prep_input_f32."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(
        timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    embedding = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        embedding = torch.cat([embedding, torch.zeros_like(embedding[:, :1])], dim=-1)
    return embedding


class _DNNForward(torch.autograd.Function):
    """model(x, t): plain denoiser forward with parameter gradients."""

    @staticmethod
    def forward(ctx, model, x, timesteps, drop_mask, *params):
        eng = model.engine
        out = eng.forward_plain(x, timesteps, training=model.training, drop_mask=drop_mask)
        ctx.eng, ctx.version = eng, eng.version
        return out

    @staticmethod
    def backward(ctx, gout):
        eng = ctx.eng
        if ctx.version != eng.version:
            raise RuntimeError("gdmcf_amd.DNN: activations were overwritten by a later forward; "
                               "call backward before the next forward")
        grads = eng.backward_plain(gout)
        return (None, None, None, None, *grads)


class DNN(nn.Module):
    """A deep neural network for the reverse diffusion process (drop-in for the reference DNN)."""

    def __init__(self, in_dims, out_dims, emb_size, time_type="cat", norm=False, dropout=0.5, gemm_dtype="f32"):
        super().__init__()
        if gemm_dtype not in ("f32", "bf16", "f32x3"):
            raise ValueError("Unimplemented GEMM input precision %s" % gemm_dtype)
        self.gemm_dtype = gemm_dtype
        self.in_dims = list(in_dims)
        self.out_dims = list(out_dims)
        assert out_dims[0] == in_dims[-1], "In and out dimensions must equal to each other."
        self.time_type = time_type
        self.time_emb_dim = emb_size
        self.norm = norm
        self.emb_layer = nn.Linear(self.time_emb_dim, self.time_emb_dim)
        if self.time_type == "cat":
            in_dims_temp = [self.in_dims[0] + self.time_emb_dim] + self.in_dims[1:]
        else:
            raise ValueError("Unimplemented timestep embedding type %s" % self.time_type)
        out_dims_temp = self.out_dims
        self.in_layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(in_dims_temp[:-1], in_dims_temp[1:])])
        self.out_layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(out_dims_temp[:-1], out_dims_temp[1:])])
        self.drop = nn.Dropout(dropout)  # holds p; the mask is applied inside the HIP input kernel
        self.init_weights()
        self._engine = None

    def init_weights(self):
        for layer in list(self.in_layers) + list(self.out_layers) + [self.emb_layer]:
            fan_out, fan_in = layer.weight.size()
            std = np.sqrt(2.0 / (fan_in + fan_out))
            layer.weight.data.normal_(0.0, std)
            layer.bias.data.normal_(0.0, 0.001)

    # -- engine -----------------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            self._engine = DenoiserEngine(self)
        return self._engine

    def __getstate__(self):  # torch.save(model) (reference main.py:375): drop device workspaces
        state = self.__dict__.copy()
        state["_engine"] = None
        return state

    def layer_list(self):
        """[(weight, bias, act)] in execution order; act 1 = tanh, 0 = none (reference :79-86)."""
        layers = [(l.weight, l.bias, 1) for l in self.in_layers]
        n_out = len(self.out_layers)
        layers += [(l.weight, l.bias, 1 if i != n_out - 1 else 0) for i, l in enumerate(self.out_layers)]
        return layers

    def param_list(self):
        return list(self.parameters())

    def forward(self, x, timesteps, drop_mask=None):
        _lib.require_gpu(x, "DNN input")
        return _DNNForward.apply(self, x, timesteps, drop_mask, *self.param_list())
