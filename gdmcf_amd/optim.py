"""Fused multi-tensor AdamW (drop-in for torch.optim.AdamW as constructed at reference main.py:258).

One HIP launch (gdmcf_adamw_f32) updates every parameter: 28 B of HBM traffic per element
(read p, g, m, v; write p, m, v).  Same update rule and the same state_dict layout
(`step`, `exp_avg`, `exp_avg_sq`) as torch.optim.AdamW (amsgrad / maximize are not supported).
"""
import os

import torch

from . import _lib

_BLOCK = 4096  # elements per workgroup; must match ADAM_BLOCK_ELEMS in kernels_misc.hip
_ROW_ALIGN = 32  # floats: fuse_into_backward seats the rows of a fused weight (and of its moments) on 128-byte lines


def _seat_rows(t, ld):
    """`t` (2-D, unit column stride) with its values in storage whose row stride is `ld` elements (== columns: contiguous);
    the elements between the rows are zero and never written afterwards."""
    n, k = t.shape
    if t.stride(1) == 1 and t.stride(0) == ld and t.storage_offset() == 0:
        return t
    buf = torch.zeros(n, ld, dtype=t.dtype, device=t.device)
    out = buf[:, :k] if ld != k else buf
    out.copy_(t)
    return out


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("FusedAdamW: amsgrad is not supported")
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        # the remaining keys are torch.optim.AdamW's own (fixed here): with them a state_dict of this optimiser loads into
        # torch.optim.AdamW with the same meaning -- without `decoupled_weight_decay` torch would fall back to Adam's L2 decay
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=True))
        self._tables = {}
        self.grad_scale = 1.0
        self._fused_ids = set()

    # -- optimiser-in-backward (single GPU, opt-in) -------------------------------------------------------------
    def fuse_into_backward(self, model, min_numel=1 << 20, align_rows=None):
        """Update the large 2-D weights of `model` (a gdmcf_amd.DNN) inside the epilogue of their weight-gradient
        GEMM instead of in step(): the gradient tile never leaves the MFMA accumulators, the separate AdamW pass over
        those tensors disappears (32 -> 24 B/param of HBM traffic).  The update rule and the resulting weights /
        moments are the same as step()'s.  Consequences: `.grad` of those weights stays None, exactly one
        backward per step() (no gradient accumulation), not for data parallel (gradients must be all-reduced
        first; DataParallelStep switches it off).  Returns self.

        Layout (align_rows, default on; GDMCF_ALIGN_ROWS=0 switches it off): the optimiser stream inside the product reads
        and writes W / exp_avg / exp_avg_sq in pieces of 256 B per row, and PyTorch's contiguous [out, in] weight starts its
        rows wherever `in` puts them (Yelp: 4 000 B and 137 620 B apart): three of four pieces then straddle three 128-byte
        lines instead of covering two, shared with the neighbouring tiles' pieces.  A fused weight is therefore RE-SEATED:
        same Parameter object, same shape and values, `.data` a view of storage whose row stride is the next multiple of
        32 floats (tools/ld_probe.py: the two Yelp products 0.291 -> 0.279 and 0.326 -> 0.285 ms).  Everything here takes
        leading dimensions, state_dict / load_state_dict / torch.save go through the view; what does need contiguous
        weights (this optimiser's separate pass, collectives) gets them back from unfuse() or another call of this method."""
        if align_rows is None:
            align_rows = os.environ.get("GDMCF_ALIGN_ROWS", "1") != "0"
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        self._fused_ids = {id(w) for (w, _, _) in model.layer_list() if w.numel() >= min_numel and id(w) in mine}
        with torch.no_grad():
            for (w, _, _) in model.layer_list():
                if id(w) not in mine or w.dim() != 2:
                    continue
                k = w.shape[1]
                ld = (k + _ROW_ALIGN - 1) // _ROW_ALIGN * _ROW_ALIGN if (id(w) in self._fused_ids and align_rows) else k
                self._seat(w, ld)
        model.engine.fused_opt = self if self._fused_ids else None
        return self

    def unfuse(self, model):
        """Back to the separate pass: nothing is updated inside the backward any more and every weight is contiguous again."""
        return self.fuse_into_backward(model, min_numel=1 << 62)

    def _seat(self, p, ld):
        if p.stride(0) != ld or p.stride(1) != 1:
            p.data = _seat_rows(p.data, ld)
        st = self.state.get(p)
        if st:
            for key in ("exp_avg", "exp_avg_sq"):
                if key in st and (st[key].stride(0) != ld or st[key].stride(1) != 1):
                    st[key] = _seat_rows(st[key], ld)

    @staticmethod
    def _zeros_seated_like(p):
        if p.dim() == 2 and p.stride(0) != p.shape[1]:
            return torch.zeros(p.shape[0], p.stride(0), dtype=p.dtype, device=p.device)[:, :p.shape[1]]
        return torch.zeros_like(p, memory_format=torch.preserve_format)

    def fused_state(self, p):
        """Called by the engine during backward: optimiser state + scalars for the coming step of `p`, or None."""
        if id(p) not in self._fused_ids:
            return None
        group = next(g for g in self.param_groups if any(q is p for q in g["params"]))
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0
            st["exp_avg"] = self._zeros_seated_like(p)
            st["exp_avg_sq"] = self._zeros_seated_like(p)
        elif st["exp_avg"].stride() != p.stride() or st["exp_avg_sq"].stride() != p.stride():
            self._seat(p, p.stride(0))  # (moments loaded from a checkpoint, or the module moved: one leading dimension for all three)
        if st.get("_fused_pending"):
            raise RuntimeError("FusedAdamW(fuse_into_backward): two backward passes without step() in between")
        st["_fused_pending"] = True
        b1, b2 = group["betas"]
        return dict(exp_avg=st["exp_avg"], exp_avg_sq=st["exp_avg_sq"], lr=float(group["lr"]), beta1=float(b1),
                    beta2=float(b2), eps=float(group["eps"]), weight_decay=float(group["weight_decay"]),
                    step=int(st["step"]) + 1, grad_scale=float(self.grad_scale))

    def _table(self, gi, plist):
        """Device tables: [n][6] = (p, g, m, v, numel, first_block) and, when any parameter has a registered bf16
        shadow (bf16 GEMM mode), [n][3] = (shadow, cols, shadow ld); re-uploaded only when a pointer moved."""
        rows, srows, blk = [], [], 0
        for p in plist:
            st = self.state[p]
            rows.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                         p.numel(), blk))
            info = _lib.shadow_info(p.data_ptr()) if p.dim() == 2 else None
            if info is not None and (info[1], info[2]) == tuple(p.shape) and p.numel() < 2 ** 32:
                srows.append((info[0], info[2], info[3]))
            else:
                srows.append((0, 1, 0))
            blk += (p.numel() + _BLOCK - 1) // _BLOCK
        key = (tuple(rows), tuple(srows))
        cache = self._tables.setdefault(gi, {})
        hit = cache.get(key)
        if hit is None:
            # gradients are fresh tensors every step; the caching allocator cycles through a handful of
            # addresses, so after a few steps every combination is resident and no H2D copy happens
            if len(cache) >= 16:
                cache.clear()
            dev = plist[0].device
            stab = torch.tensor(srows, dtype=torch.int64).to(dev) if any(r[0] for r in srows) else None
            hit = (torch.tensor(rows, dtype=torch.int64).to(dev), blk, stab, [r[0] != 0 for r in srows])
            cache[key] = hit
        return hit

    def _launch(self, gi, group, plist, step):
        lib = _lib.load()
        for p in plist:
            _lib.require_gpu(p, "FusedAdamW parameter")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("FusedAdamW: parameters must be contiguous float32 (a weight seated on aligned rows by "
                                   "fuse_into_backward is not: unfuse(model) first)")
            if p.grad.is_sparse:
                raise RuntimeError("FusedAdamW does not support sparse gradients")
            if not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                p.grad = p.grad.float().contiguous()
        table, nblk, stab, shadowed = self._table(gi, plist)
        b1, b2 = group["betas"]
        if stab is None:
            _lib.check(lib.gdmcf_adamw_f32(table.data_ptr(), len(plist), nblk, group["lr"], b1, b2, group["eps"],
                                           group["weight_decay"], step, float(self.grad_scale), _lib.stream_ptr()))
        else:
            _lib.check(lib.gdmcf_adamw_bf16s_f32(table.data_ptr(), stab.data_ptr(), len(plist), nblk, group["lr"], b1, b2,
                                                 group["eps"], group["weight_decay"], step, float(self.grad_scale),
                                                 _lib.stream_ptr()))
        # The kernel wrote through raw pointers.  Parameters whose bf16 shadow was refreshed by this very launch keep
        # their version (the engine's version-keyed shadow check then sees nothing to redo); all others are bumped so
        # that autograd and version-keyed caches notice the in-place update.
        for p, sh in zip(plist, shadowed):
            if not sh:
                torch.autograd.graph.increment_version(p)

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif p.dim() == 2 and (st["exp_avg"].stride() != p.stride() or st["exp_avg_sq"].stride() != p.stride()):
            # moments loaded from a checkpoint written while the weight sat on aligned rows (or the other way round): the kernels
            # address W, exp_avg and exp_avg_sq with ONE leading dimension
            self._seat(p, p.stride(0))
        return st

    @torch.no_grad()
    def step_rows(self, param, grad_rows, row0):
        """Sharded-optimiser building block (parallel.DataParallelStep(shard_optimizer=True)): apply this step's update
        to rows [row0, row0 + grad_rows.shape[0]) of the 2-D `param` only, with `grad_rows` (contiguous, already
        reduced over the ranks) as their gradient.  The moments are full-size tensors of which a rank keeps just its
        rows current.  May be called several times per step for disjoint row ranges; the following step() skips the
        tensor (as after step_subset)."""
        if param.dim() != 2 or not param.is_contiguous() or grad_rows.dim() != 2 or grad_rows.shape[1] != param.shape[1]:
            raise RuntimeError("FusedAdamW.step_rows: 2-D contiguous parameter and matching gradient rows expected")
        n = int(grad_rows.shape[0])
        if n == 0:
            return
        gi, group = next((i, g) for i, g in enumerate(self.param_groups) if any(q is param for q in g["params"]))
        st = self._init_state(param)
        g = grad_rows if grad_rows.is_contiguous() and grad_rows.dtype == torch.float32 else grad_rows.float().contiguous()
        C, off = param.shape[1], row0 * param.shape[1] * 4
        row = (param.data_ptr() + off, g.data_ptr(), st["exp_avg"].data_ptr() + off, st["exp_avg_sq"].data_ptr() + off, n * C, 0)
        cache = self.__dict__.setdefault("_row_cache", {})
        table = cache.get(row)  # gradient shards cycle through a few allocator addresses: no H2D copy once warm
        if table is None:
            if len(cache) >= 32:
                cache.clear()
            table = cache[row] = torch.tensor([row], dtype=torch.int64).to(param.device)
        lib = _lib.load()
        b1, b2 = group["betas"]
        _lib.check(lib.gdmcf_adamw_f32(table.data_ptr(), 1, (n * C + _BLOCK - 1) // _BLOCK, group["lr"], b1, b2, group["eps"],
                                       group["weight_decay"], int(st["step"]) + 1, float(self.grad_scale), _lib.stream_ptr()))
        self._row_tables = getattr(self, "_row_tables", [])
        self._row_tables.append((table, g))  # alive until the stream has consumed them (dropped at the next step())
        st["_fused_pending"] = True
        torch.autograd.graph.increment_version(param)

    @torch.no_grad()
    def step_subset(self, params):
        """Apply this step's update to `params` now; the following step() handles the remaining tensors and skips
        these.  Lets a data-parallel driver update a tensor as soon as its all-reduce has finished while other
        gradients are still on the wire."""
        params = [p for p in params if p.grad is not None]
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if any(p is q for q in params)]
            if not plist:
                continue
            sts = [self._init_state(p) for p in plist]
            if any(st.get("_fused_pending") for st in sts):
                raise RuntimeError("FusedAdamW.step_subset: tensor already updated in this step")
            self._launch(gi, group, plist, int(sts[0]["step"]) + 1)
            for st in sts:
                st["_fused_pending"] = True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._row_tables = []  # row-shard tables of the previous step: their kernels are long enqueued behind us
        for gi, group in enumerate(self.param_groups):
            done = set()
            for p in group["params"]:  # tensors already updated (inside the backward pass or by step_subset)
                st = self.state.get(p)
                if st and st.get("_fused_pending"):
                    st["step"] = int(st["step"]) + 1
                    st["_fused_pending"] = False
                    done.add(id(p))
            plist = [p for p in group["params"] if p.grad is not None and id(p) not in done]
            if not plist:
                continue
            steps = {int(self._init_state(p)["step"]) for p in plist}
            if len(steps) != 1:
                raise RuntimeError("FusedAdamW: parameters of one group must share the step count")
            step = steps.pop() + 1
            self._launch(gi, group, plist, step)
            for p in plist:
                self.state[p]["step"] = step
        return loss
