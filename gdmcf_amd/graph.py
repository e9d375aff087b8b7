"""The training step as ONE hipGraph (MI355X-native: capture the launch-bound loop instead of tracing it).

`GraphedTrainStep(diffusion, model, optimizer, csr, batch_size)` captures the body of the reference's loop (main.py:345-351:
zero_grad -> training_losses -> mean -> backward -> AdamW.step; 18 kernel launches at the Yelp shape) once and replays it per
batch: the host then enqueues one graph launch and one 3 KB copy of the batch's row ids per step instead of ~150 Python /
ctypes calls.  A replay runs the SAME launches with the SAME arguments, so everything that changes from step to step lives
in device memory:
  * the batch: rows `ids` of a device-resident CSR matrix (data_utils.CsrBatch over a fixed id buffer that is overwritten
    before every replay);
  * the Philox offsets of the input builder and of the timestep sampler, and the AdamW bias corrections: the library's graph
    step state (include/gdmcf_hip.h: gdmcf_graph_state_*), advanced by one tick kernel at the head of the graph; the
    corrections come from a host-computed table, so the updates are bit-identical to the eager path;
  * the Lt-history (already on the device), gradients (persistent buffers: DenoiserEngine.static_grads), the loss.
Plain DNN denoiser, in-kernel Philox randomness, x0 target (what bench.py times); other configurations use DataParallelStep.
Data parallel (world > 1, backend "nccl" = RCCL; or `force_exchange=True` in a one-rank group, the rehearsal of that path on a
one-GPU box): the captured body is DataParallelStep's -- the gradient all-reduces and the small float64 exchange are RCCL
calls on torch's collective stream, forked from and joined to the capturing stream by events, so they become nodes of the
same graph and the host enqueues ONE launch per step on every rank (the all-reduce exchange; the sharded optimiser defers
its all-gathers into the NEXT step and stays eager).  If the capture is refused (an RCCL build that cannot be captured), the
step keeps running eagerly in the same process -- `capture_error` says why -- never a re-exec.  The first `warmup` calls run eagerly (same kernels, same device state: they are ordinary training steps),
the next call captures and replays.  `close()` (or leaving the `with` block) writes the device counters back into the
host-side counters (optimizer step counts, Philox offsets) so that eager steps, checkpoints and resumes continue seamlessly.
"""
import ctypes

import torch

from . import _lib
from .DNN import DNN
from .data_utils import CsrBatch
from .gaussian_diffusion import ModelMeanType
from .parallel import DataParallelStep


class GraphedTrainStep:
    def __init__(self, diffusion, model, optimizer, csr, batch_size, reweight=True, warmup=3, table_steps=8192, group=None,
                 force_exchange=False):
        dist = torch.distributed
        multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_exchange)
        # a gradient exchange can only be captured when its collectives are device work (backend "nccl" = RCCL): host-staged
        # ones (gloo rehearsals) are not -- the step then runs eagerly in this process and `capture_error` says why
        self._no_capture = None
        if multi and dist.get_backend(group) != "nccl":
            self._no_capture = ("NotImplementedError: a captured gradient exchange needs the nccl (RCCL) backend; backend "
                                f"'{dist.get_backend(group)}' stages its collectives through the host")
        if not isinstance(model, DNN) or model.norm or diffusion.mean_type != ModelMeanType.START_X:
            raise NotImplementedError("GraphedTrainStep: plain DNN denoiser without F.normalize, x0 target")
        if getattr(diffusion, "CatOneHot", False) or diffusion.rng != "philox" or diffusion.noise_scale == 0.0:
            raise NotImplementedError("GraphedTrainStep: continuous diffusion with in-kernel Philox randomness")
        if len(optimizer.param_groups) != 1 or not hasattr(optimizer, "grad_scale"):
            raise NotImplementedError("GraphedTrainStep: a gdmcf_amd.FusedAdamW, one parameter group")
        # (FusedAdamW.fuse_into_backward is fine since round 4: the weight-gradient products with the optimiser inside read this
        # step's AdamW scalars from the bound step state, like the stand-alone AdamW kernel)
        if csr.values is not None:
            raise NotImplementedError("GraphedTrainStep: interaction values must all be 1 (rows stay sparse)")
        self.diffusion, self.model, self.optimizer, self.reweight = diffusion, model, optimizer, reweight
        self.lib = _lib.load()
        dev = csr.device
        self.ids = torch.zeros(batch_size, dtype=torch.int64, device=dev)
        self.batch = CsrBatch(csr, self.ids)
        assert self.batch.row_ids.data_ptr() == self.ids.data_ptr()
        self.eng = model.engine
        self.eng.static_grads = True
        # (constructed BEFORE the AdamW table is uploaded: it sets optimizer.grad_scale = 1 / world)
        self.step = DataParallelStep(diffusion, model, optimizer, group=group, force_exchange=bool(force_exchange) and multi)
        self.warmup, self.calls, self.graph, self.loss = int(warmup), 0, None, None
        self.capture_error = None
        self.table_steps = int(table_steps)
        self._state = None
        self._upload_state()

    # -- device step state ---------------------------------------------------------------------------------------------
    def _host_counters(self):
        steps = {int(st.get("step", 0)) for st in self.optimizer.state.values() if "step" in st} or {0}
        if len(steps) != 1:
            raise RuntimeError("GraphedTrainStep: all parameters must share the optimizer step count")
        return int(self.eng.offset), int(getattr(self.diffusion, "_ts_calls", 0)), steps.pop()

    def _upload_state(self, counters=None):
        lib, dev = self.lib, self.ids.device
        prep, ts, step = counters if counters is not None else self._host_counters()
        g = self.optimizer.param_groups[0]
        self._hyper = self._hyper_now()
        hb, sb = lib.gdmcf_adam_hyper_bytes(), lib.gdmcf_graph_state_bytes()
        host_tab = (ctypes.c_ubyte * (hb * self.table_steps))()
        _lib.check(lib.gdmcf_adam_hyper_fill(host_tab, self.table_steps, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                             g["weight_decay"], step + 1, float(self.optimizer.grad_scale)))
        self._table = torch.frombuffer(host_tab, dtype=torch.uint8).clone().to(dev)
        host_st = (ctypes.c_ubyte * sb)()
        _lib.check(lib.gdmcf_graph_state_init(host_st, prep, ts, step, step + 1, self.table_steps, self._table.data_ptr()))
        new = torch.frombuffer(host_st, dtype=torch.uint8).clone().to(dev)
        if getattr(self, "_state", None) is None:
            self._state = new
        else:
            self._state.copy_(new)  # same address: the captured graph keeps pointing at it
        self._table_end = step + self.table_steps  # last optimiser step the table covers
        self._dev_step = step                      # host mirror of the device's optimiser step count

    def _hyper_now(self):
        g = self.optimizer.param_groups[0]
        return (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                float(self.optimizer.grad_scale))

    def _read_state(self):
        torch.cuda.synchronize()
        raw = self._state.cpu().numpy().tobytes()
        prep, ts = int.from_bytes(raw[0:8], "little"), int.from_bytes(raw[8:16], "little")
        step = int.from_bytes(raw[16:24], "little", signed=True)
        return prep, ts, step

    # -- one training step ---------------------------------------------------------------------------------------------
    def _body(self):
        _lib.check(self.lib.gdmcf_graph_state_tick(self._state.data_ptr(), _lib.stream_ptr()))
        return self.step(self.batch, self.reweight)

    def __call__(self, row_ids):
        """Trains on rows `row_ids` (int64 tensor of batch_size ids, any device) of the CSR matrix; returns the mean loss."""
        if row_ids.numel() != self.ids.numel():
            raise ValueError("GraphedTrainStep: every batch must have the captured batch size")
        self.ids.copy_(row_ids, non_blocking=True)
        self.calls += 1
        # the AdamW scalars of the coming steps sit in a device table computed from the optimizer's hyper-parameters: a changed
        # learning rate (scheduler, manual edit) or an exhausted table means a new table -- same address, the graph stays valid
        if self._hyper_now() != self._hyper or self._dev_step + 1 > self._table_end:
            self._upload_state(self._read_state())
        if self.graph is None or self.graph is False:
            self.lib.gdmcf_graph_state_bind(self._state.data_ptr())
            try:
                if self.calls <= self.warmup or self.graph is False:
                    self._dev_step += 1
                    return self._body().clone()
                if self._no_capture is not None:
                    self.graph, self.capture_error = False, self._no_capture
                    self._dev_step += 1
                    return self._body().clone()
                self.lib.gdmcf_prof_enable(0)  # event records cannot be captured
                g = torch.cuda.CUDAGraph()
                try:
                    # RCCL's watchdog thread polls the events of earlier (eager) collectives: under the default "global"
                    # capture mode such a query from ANOTHER thread is an error that kills the process; thread-local mode
                    # only polices the capturing thread.  Nothing of the eager steps may still be in flight either way.
                    torch.cuda.synchronize()
                    if self.step.exchange:
                        # ... and the watchdog (it wakes every 100 ms) must have seen them complete and dropped them from its
                        # list: a poll of one of those events that lands inside the capture aborted the process in about one
                        # run in twenty (round 4, tests/test_gpu_dp.py under load) although the mode is thread-local
                        import time
                        time.sleep(0.35)
                    with torch.cuda.graph(g, capture_error_mode="thread_local" if self.step.exchange else "global"):
                        self.loss = self._body()
                    self.graph = g
                except Exception as exc:  # e.g. a collective library that refuses stream capture: stay eager, same process
                    self.graph, self.capture_error = False, f"{type(exc).__name__}: {exc}"[:300]
                    torch.cuda.synchronize()
                    self._dev_step += 1
                    return self._body().clone()
            finally:
                self.lib.gdmcf_graph_state_bind(None)
        self.graph.replay()
        self._dev_step += 1
        return self.loss.clone()

    # -- hand the counters back to the host-side objects ---------------------------------------------------------------------
    def close(self):
        """Write the device counters back (optimizer step counts, Philox offsets) and return the engine to fresh gradient
        tensors: eager steps, checkpoints and resumes then continue exactly where the graph stopped."""
        if self._state is None:
            return
        prep, ts, step = self._read_state()
        self.eng.offset = prep
        if self.reweight:  # (the sampler's Philox position only moves when importance sampling draws timesteps)
            self.diffusion._ts_calls = ts
        for st in self.optimizer.state.values():
            if "step" in st:
                st["step"] = step
        self.eng.static_grads = False
        self.graph, self._state = None, None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
