"""Data parallelism for the training step: one process per GPU, torch.distributed (backend "nccl"
is RCCL on ROCm, over xGMI); "gloo" on CPU for tests.

User rows are independent through q_sample, the denoiser forward and the per-row loss, so a batch
shards over ranks with exactly one exchange per step:
  * all-reduce(SUM) of the parameter gradients, scaled by 1/world in the fused AdamW
    (rank r holds rows [r*B, (r+1)*B) of the global batch; grad of the global mean = mean of
    the local-mean grads);
  * an all-gather of (ts, unscaled loss) -- 16 B per row -- so every rank replays the identical
    order-dependent Lt-history FIFO update (reference gaussian_diffusion.py:355-368) over the
    global batch in rank order.
The reference has no distributed path; this is new, MI355X-first design.
"""
import functools

import torch
import torch.distributed as dist


def _host_staged(group):
    """gloo has no device collectives on ROCm: stage through the host (rehearsal/tests only; the
    production backend is "nccl" = RCCL, device to device over xGMI)."""
    return dist.get_backend(group) == "gloo"


def _all_reduce(t, group, async_op=False):
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def all_gather_rows(local, n_total, group=None):
    """[rows_per_rank, d] blocks of every rank, concatenated in rank order and cut to n_total rows (row-sharded
    LightGCN propagation: every rank needs the full embedding table of the previous layer)."""
    world = dist.get_world_size(group)
    if local.is_cuda and _host_staged(group):
        h = local.cpu()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h, group=group)
        return torch.cat(parts)[:n_total].to(local.device)
    out = torch.empty(world * local.shape[0], local.shape[1], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:n_total]


def reduce_scatter_rows(rows, group=None):
    """SUM over ranks of `rows` [world * n, C] (contiguous), returning this rank's block [n, C] and a work handle (None
    when the collective already completed)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = rows.shape[0] // world
    if rows.is_cuda and _host_staged(group):  # gloo rehearsal: all-reduce on the host, keep the own block
        h = rows.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        return h[rank * n:(rank + 1) * n].to(rows.device), None
    if _host_staged(group):
        h = rows.clone()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        return h[rank * n:(rank + 1) * n].contiguous(), None
    out = torch.empty(n, rows.shape[1], dtype=rows.dtype, device=rows.device)
    return out, dist.reduce_scatter_tensor(out, rows, op=dist.ReduceOp.SUM, group=group, async_op=True)


def all_gather_rows_inplace(full, group=None):
    """`full` [world * n, C] (contiguous): every rank has written its own block of n rows; fetch the other ranks'
    blocks in place."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = full.shape[0] // world
    mine = full[rank * n:(rank + 1) * n]
    if _host_staged(group):
        h = mine.cpu() if mine.is_cuda else mine.clone()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h, group=group)
        full.copy_(torch.cat(parts).to(full.device))
        return None
    return dist.all_gather_into_tensor(full, mine, group=group, async_op=True)


def allreduce_grads(params, group=None, bucket_bytes=256 << 20, force=False):
    """In-place SUM all-reduce of .grad over ranks.  Large tensors go alone (no copy); small ones are
    coalesced into one flat bucket.  xGMI is point-to-point, so few, large messages are preferred."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    small, handles = [], []
    for p in params:
        if p.grad is None:
            continue
        g = p.grad
        if g.numel() * g.element_size() >= (1 << 20):
            handles.append(_all_reduce(g, group, async_op=True))
        else:
            small.append(g)
    if small:
        flat = torch.cat([g.reshape(-1) for g in small])
        _all_reduce(flat, group)
        off = 0
        for g in small:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    for h in handles:
        if h is not None:
            h.wait()


def gather_history_inputs(ts, loss_unscaled, group=None, force=False):
    """All-gather (ts, unscaled per-row loss) in rank order -> tensors of the GLOBAL batch."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return ts, loss_unscaled
    world = dist.get_world_size(group)
    dev = ts.device
    if ts.is_cuda and _host_staged(group):
        ts, loss_unscaled = ts.cpu(), loss_unscaled.cpu()
    ts_all = [torch.empty_like(ts) for _ in range(world)]
    lu_all = [torch.empty_like(loss_unscaled) for _ in range(world)]
    dist.all_gather(ts_all, ts.contiguous(), group=group)
    dist.all_gather(lu_all, loss_unscaled.contiguous(), group=group)
    return torch.cat(ts_all).to(dev), torch.cat(lu_all).to(dev)


def broadcast_parameters(model, group=None, src=0, force=False):
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        for p in model.parameters():
            if p.is_cuda and _host_staged(group):
                h = p.data.cpu()
                dist.broadcast(h, src=src, group=group)
                p.data.copy_(h)
            else:
                dist.broadcast(p.data, src=src, group=group)
            # writes through .data do not move the version counter; caches keyed on it (bf16 weight shadows) must see this
            torch.autograd.graph.increment_version(p)


class DataParallelStep:
    """zero_grad -> training_losses -> mean -> backward -> all-reduce -> AdamW, per rank (the body of
    reference main.py:345-351 plus the single exchange).

    Overlap: the denoiser engine hands every gradient to `_sink` as soon as its kernels are enqueued.
    Large tensors (the two 137.6 MB weight gradients at Yelp shape) start their all-reduce immediately on
    RCCL's stream, so out_layers' gradient travels over xGMI while the dh / dW1 GEMMs still run; the few
    small tensors are reduced in one flat bucket at the end.  Everything is waited for before AdamW.

    `shard_optimizer=True` (opt-in): large 2-D weights are reduce-scattered by row blocks instead of all-reduced, every
    rank runs AdamW on its 1/world of the rows only (`FusedAdamW.step_rows`) and the updated rows are all-gathered in
    place -- the same bytes on the wire as an all-reduce, 1/world of the optimiser's HBM traffic per GPU.  The moments
    of a sharded weight are then current only in the rank's own rows (`gather_optimizer_state()` before saving).
    The all-gathers are issued first-layer first and NOT waited for at the end of the step (`defer_gather`): the engine
    waits right before the first GEMM that reads each weight, so the last layer's gather travels under the next step's
    input builder and first-layer GEMMs.  Any forward through the engine and `model.state_dict()` wait by themselves;
    call `flush()` before reading `model.parameters()` directly.

    Single process (world == 1): no exchange.  `early_update=True` issues the AdamW update of a large tensor on a
    side stream the moment its gradient GEMM is enqueued (the engine then computes a layer's input gradient BEFORE
    its weight gradient, so nothing reads the old weight any more); same kernels, same values as the sequential
    order, streams joined before `optimizer.step()` finishes the remaining tensors.  Off by default: measured on
    MI355X at the Yelp shape it LOSES 3-4 % (1.76 -> 1.83 ms/step) -- the HBM-bound update and the MFMA-bound dW1
    GEMM slow each other down by more than the 0.17 ms that is hidden."""

    def __init__(self, diffusion, model, optimizer, group=None, overlap=True, early_update=False, direct_backward=True,
                 shard_optimizer=False, force_exchange=False, defer_gather=True):
        self.diffusion, self.model, self.optimizer, self.group = diffusion, model, optimizer, group
        self.direct_backward = direct_backward
        self.defer_gather = bool(defer_gather)
        self.shard_optimizer = (bool(shard_optimizer) and overlap and hasattr(optimizer, "step_rows")
                                and getattr(model.engine, "supports_grad_sink", False))
        self._sharded = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._handles, self._small = [], []
        # force_exchange: run every collective even in a group of one rank (rehearsal of the RCCL code path on a
        # single-GPU box; a one-rank SUM is the identity, so the step must equal the plain single-process step)
        self.exchange = self.world > 1 or (bool(force_exchange) and dist.is_initialized())
        if self.exchange:
            diffusion.update_history = False  # replayed below on the gathered global batch
            optimizer.grad_scale = 1.0 / self.world
            if hasattr(optimizer, "unfuse") and hasattr(model, "layer_list"):
                optimizer.unfuse(model)  # gradients must be all-reduced before the update (and collectives want contiguous weights)
            model.engine.fused_opt = None
            if hasattr(optimizer, "_fused_ids"):
                optimizer._fused_ids = set()
            broadcast_parameters(model, group, force=True)
            if overlap and getattr(model.engine, "supports_grad_sink", False):
                model.engine.grad_sink = self._sink  # (other engines: gradients are exchanged after the backward)
            if getattr(model.engine, "supports_grad_sink", False):
                model.register_state_dict_pre_hook(lambda *a, **k: self.flush())  # complete weights in checkpoints
        elif (early_update and hasattr(optimizer, "step_subset") and getattr(model.engine, "fused_opt", None) is None
              and next(model.parameters()).is_cuda):
            self._side = torch.cuda.Stream()
            self._side_busy = False
            model.engine.grad_sink = self._sink_local
            model.engine.input_grad_first = True

    def _sink(self, param, grad):
        if (self.shard_optimizer and param.grad is None and grad.dim() == 2 and grad.is_contiguous()
                and grad.numel() * grad.element_size() >= (1 << 20) and grad.shape[0] >= self.world):
            # sharded optimiser: rank r receives the summed gradient of its block of rows only (half the bytes of an
            # all-reduce on the wire), updates those rows, and the updated rows are all-gathered afterwards; the
            # R mod world leftover rows are all-reduced and updated by every rank
            n_eq = grad.shape[0] // self.world * self.world
            shard, h = reduce_scatter_rows(grad[:n_eq], self.group)
            tail = grad[n_eq:]
            h2 = _all_reduce(tail, self.group, async_op=True) if tail.numel() else None
            self._sharded.append((param, grad, shard, h, tail, h2, n_eq))
            return
        param.grad = grad if param.grad is None else param.grad.add_(grad)
        g = param.grad
        if g.numel() * g.element_size() >= (1 << 20):
            self._handles.append((param, _all_reduce(g, self.group, async_op=True)))
        else:
            self._small.append(g)

    def set_shard_optimizer(self, flag):
        """Switch between the sharded optimiser and the plain all-reduce path between steps (bench.py times both during
        warm-up and keeps the faster one).  Leaving the sharded mode first completes weights and moments on every rank."""
        flag = (bool(flag) and hasattr(self.optimizer, "step_rows") and self.exchange
                and getattr(self.model.engine, "grad_sink", None) is not None)
        if self.shard_optimizer and not flag:
            self.gather_optimizer_state()
        self.shard_optimizer = flag
        return flag

    def gather_optimizer_state(self):
        """Sharded optimiser: make exp_avg / exp_avg_sq of the sharded weights complete on every rank (checkpoints)."""
        if not (self.shard_optimizer and self.exchange):
            return
        self.flush()
        for p in self.model.parameters():
            st = self.optimizer.state.get(p)
            if st and p.dim() == 2 and p.numel() * 4 >= (1 << 20) and p.shape[0] >= self.world:
                n_eq = p.shape[0] // self.world * self.world
                for key in ("exp_avg", "exp_avg_sq"):
                    h = all_gather_rows_inplace(st[key][:n_eq], self.group)
                    if h is not None:
                        h.wait()

    def _sink_local(self, param, grad):
        param.grad = grad if param.grad is None else param.grad.add_(grad)
        if param.grad.numel() * param.grad.element_size() < (1 << 20):
            return
        self.optimizer._init_state(param)  # allocate the moments on the main stream
        done = torch.cuda.Event()
        done.record()  # everything enqueued so far: this gradient and every reader of the old weight
        self._side.wait_event(done)
        with torch.cuda.stream(self._side):
            self.optimizer.step_subset([param])
        self._side_busy = True

    def _finish_exchange(self):
        """One small float64 all-reduce carries (a) every small gradient and (b) the (ts, unscaled loss) pairs of
        all ranks: each rank writes its slice of a zero-initialised [world, B, 2] block, so SUM == all-gather in
        rank order (gdmcf_dp_pack_f64 / gdmcf_dp_unpack_f64: one kernel each side of the collective).  Returns
        (ts_all, lu_all) of the global batch."""
        import ctypes
        from . import _lib
        lib, st = _lib.load(), _lib.stream_ptr()
        d = self.diffusion
        ts, lu = d.last_ts, d.last_loss_unscaled
        B, dev = ts.numel(), ts.device
        rank = dist.get_rank(self.group)
        small = self._small
        for i, g in enumerate(small):
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise RuntimeError("DataParallelStep: small gradients must be contiguous float32")
        n = len(small)
        if n > 16:
            raise RuntimeError("DataParallelStep: more than 16 small gradient tensors")
        ptrs = (ctypes.c_void_p * max(n, 1))(*[g.data_ptr() for g in small])
        counts = (ctypes.c_int64 * max(n, 1))(*[g.numel() for g in small])
        n_small = sum(g.numel() for g in small)
        flat = torch.empty(n_small + self.world * B * 2, dtype=torch.float64, device=dev)
        ts = ts if ts.dtype == torch.int64 and ts.is_contiguous() else ts.to(torch.int64).contiguous()
        lu = lu if lu.dtype == torch.float64 and lu.is_contiguous() else lu.double().contiguous()
        _lib.check(lib.gdmcf_dp_pack_f64(ptrs, counts, n, ts.data_ptr(), lu.data_ptr(), B, rank, self.world,
                                         flat.data_ptr(), st))
        # sharded optimiser: the tiny bucket goes on the wire right behind the reduce-scatters, ahead of the all-gathers
        hflat = _all_reduce(flat, self.group, async_op=True) if self._sharded else None
        pending = []
        for param, grad, shard, h, tail, h2, n_eq in self._sharded:
            if h is not None:
                h.wait()
            nr = n_eq // self.world
            self.optimizer.step_rows(param, shard, rank * nr)
            if tail.numel():
                if h2 is not None:
                    h2.wait()
                self.optimizer.step_rows(param, tail, n_eq)
            pending.append((param, n_eq))
        # gradients arrive last-layer first, the next forward needs the first layer first: gather in reverse order
        self._gathers = [(param, all_gather_rows_inplace(param.data[:n_eq], self.group)) for param, n_eq in reversed(pending)]
        had_sharded, self._sharded = bool(self._sharded), []
        # the large all-reduces complete in launch order; every tensor but the last is updated the moment its
        # own reduction is done, so that AdamW pass overlaps the reductions still on the wire
        early = hasattr(self.optimizer, "step_subset")
        for k, (param, h) in enumerate(self._handles):
            if h is not None:
                h.wait()
            if early and k + 1 < len(self._handles):
                self.optimizer.step_subset([param])
        if not had_sharded:
            _all_reduce(flat, self.group)
        elif hflat is not None:
            hflat.wait()
        ts_all = torch.empty(self.world * B, dtype=torch.int64, device=dev)
        lu_all = torch.empty(self.world * B, dtype=torch.float64, device=dev)
        _lib.check(lib.gdmcf_dp_unpack_f64(flat.data_ptr(), ptrs, counts, n, B, self.world, ts_all.data_ptr(),
                                           lu_all.data_ptr(), st))
        self._handles, self._small = [], []
        return ts_all, lu_all

    def __call__(self, batch, reweight=True, **rand):
        from . import _lib
        self.optimizer.zero_grad()
        eng = getattr(self.model, "engine", None)
        if self.direct_backward and eng is not None:
            # mean reduction has the constant upstream gradient 1/B: run the backward directly instead of through an
            # autograd graph (same kernels on the same values; saves four tiny elementwise launches per step)
            with torch.no_grad():
                losses = self.diffusion.training_losses(self.model, batch, reweight, **rand)
            loss = getattr(eng, "last_loss_mean", None)  # written by the loss tail kernel (fixed-order float64 mean)
            if loss is None:
                loss = losses["loss"].mean()
            grads = eng.train_backward(1.0 / losses["loss"].numel())
            for p, g in zip(self.model.param_list(), grads):
                if g is not None:  # None: already handed to the gradient sink
                    p.grad = g if p.grad is None else p.grad.add_(g)
        else:
            losses = self.diffusion.training_losses(self.model, batch, reweight, **rand)
            loss = losses["loss"].mean()
            loss.backward()
            if getattr(eng, "last_loss_mean", None) is not None:
                loss = eng.last_loss_mean  # report the same (fixed-order) mean on every path
        if self.exchange:
            d = self.diffusion
            if getattr(self.model.engine, "grad_sink", None) is not None:
                ts_all, lu_all = self._finish_exchange()
            else:
                allreduce_grads(self.model.parameters(), self.group, force=True)
                ts_all, lu_all = gather_history_inputs(d.last_ts, d.last_loss_unscaled, self.group, force=True)
            _lib.check(_lib.load().gdmcf_lt_history_update(ts_all.data_ptr(), lu_all.data_ptr(), ts_all.numel(),
                                                           d.steps, d.history_num_per_term, d.Lt_history.data_ptr(),
                                                           d.Lt_count.data_ptr(), _lib.stream_ptr()))
        if getattr(self, "_side_busy", False):
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_busy = False
        self.optimizer.step()
        for param, h in getattr(self, "_gathers", []):  # updated row blocks of the other ranks
            if self.defer_gather and eng is not None and hasattr(eng, "weight_waiters"):
                # still on the wire: the engine waits right before the first GEMM of the next forward that reads it
                eng.weight_waiters[id(param)] = functools.partial(self._arrived, param, h)
            else:
                self._arrived(param, h)
        self._gathers = []
        return loss.detach()

    @staticmethod
    def _arrived(param, h):
        if h is not None:
            h.wait()
        torch.autograd.graph.increment_version(param)

    def flush(self):
        """Wait (on the current stream) for every all-gather of updated weight rows still in flight.  The engine does
        this by itself before any forward; call it before reading the parameters directly (checkpoints, copies)."""
        eng = getattr(self.model, "engine", None)
        if eng is not None and hasattr(eng, "flush_weight_waiters"):
            eng.flush_weight_waiters()
