"""ctypes binding of libgdmcf_hip.so (C ABI: include/gdmcf_hip.h).

The product path has NO CPU fallback: if the HIP library is missing, or a kernel is asked to run
without a GPU, this module raises -- it never routes through PyTorch eager or the oracle.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgdmcf_hip.so")

GDMCF_OK, E_SHAPE, E_ARG, E_UNSUPPORTED, E_HIP, E_WORKSPACE = 0, -1, -2, -3, -4, -5
N_TABLES = 13
TABLE_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2",
)

P = c_void_p
_SIGNATURES = {
    "gdmcf_version": (c_int, []),
    "gdmcf_debug_last_gemm": (c_int, []),
    "gdmcf_last_error": (c_char_p, []),
    "gdmcf_device_info": (c_int, [P, P, c_char_p, c_int]),
    "gdmcf_prof_enable": (c_int, [c_int]),
    "gdmcf_prof_collect": (c_int, [c_int, P, P, P]),
    "gdmcf_schedule_build": (c_int, [c_int, c_double, c_double, c_double, c_int, c_int, P]),
    "gdmcf_densify_rows_f32": (c_int, [P, P, P, P, c_int, c_int, P, c_int64, P]),
    "gdmcf_dnn_prep_input_f32": (c_int, [P, c_int64, P, P, P, c_int, P, c_int64, c_int, P, c_int64, c_float, c_uint64,
                                         c_uint64, c_int, P, P, c_int, c_int, c_int, P, c_int64, P, c_int64, P, P, P]),
    "gdmcf_dnn_emb_cols_f32": (c_int, [P, P, P, c_int, c_int, c_int, P, c_int64, P, P]),
    "gdmcf_topn_metrics_f64": (c_int, [P, c_int64, c_int, P, P, P, c_int, P, P]),
    "gdmcf_gemm_precision": (c_int, [c_int]),
    "gdmcf_bf16_shadow_set": (c_int, [P, P, c_int64, c_int64, c_int64]),
    "gdmcf_bf16_shadow_clear": (c_int, [P]),
    "gdmcf_bf16_shadow_get": (P, [P]),
    "gdmcf_bf16_shadow_info": (c_int, [P, P, P, P, P]),
    "gdmcf_bf16_shadow_sync": (c_int, [P, c_int64, P]),
    "gdmcf_linear_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gdmcf_linear_fwd_f32": (c_int, [P, c_int64, P, c_int64, P, c_int, c_int, c_int, c_int, P, c_int64, P, c_size_t, P]),
    "gdmcf_linear_fwd_wt_f32": (c_int, [P, c_int64, P, c_int64, P, c_int, c_int, c_int, c_int, P, c_int64, P, c_size_t, P]),
    "gdmcf_loss_tiles": (c_int, [c_int]),
    "gdmcf_linear_loss_fwd_f32": (c_int, [P, c_int64, P, c_int64, P, P, c_int64, P, c_int, c_int, c_int, P, c_int64, P,
                                          c_int64, P, P, P]),
    "gdmcf_linear_posterior_fwd_f32": (c_int, [P, c_int64, P, c_int64, P, P, c_int64, P, P, P, P, P, P, c_int64, c_int,
                                               c_int, c_int, P, c_int64, P, c_int64, P]),
    "gdmcf_linear_bwd_input_f32": (c_int, [P, c_int64, P, c_int64, P, P, c_int64, c_int, c_int, c_int, c_int, P,
                                           c_int64, P, c_size_t, P]),
    "gdmcf_linear_bwd_weight_f32": (c_int, [P, c_int64, P, c_int64, P, c_int, c_int, c_int, c_int, P, c_int64, P, c_int, P]),
    "gdmcf_linear_bwd_weight_adamw_f32": (c_int, [P, c_int64, P, c_int64, P, c_int, c_int, c_int, c_int, P, c_int64, P, P, P, c_float,
                                                  c_float, c_float, c_float, c_float, c_int, c_float, P]),
    "gdmcf_rowscale_f32": (c_int, [P, c_int64, P, c_int, c_int, P, c_int64, P]),
    "gdmcf_emb_bwd_f32": (c_int, [P, c_int64, P, c_int64, c_int, c_int, P, c_int, c_int, P, P, P, P]),
    "gdmcf_row_loss_finish_f64": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, P, P, c_int, P, P, P, P]),
    "gdmcf_lt_history_update": (c_int, [P, P, c_int, c_int, c_int, P, P, P]),
    "gdmcf_onehot_noise_f32": (c_int, [P, c_int64, P, c_int, c_int, c_float, P, c_int64, c_uint64, c_uint64, P, c_int64, P,
                                     c_int64, P]),
    "gdmcf_row_norms_f32": (c_int, [P, c_int64, c_int, c_int, P, P, P]),
    "gdmcf_normalize_rows_bwd_f32": (c_int, [P, c_int64, P, c_int64, P, c_int, c_int, P, c_int64, P]),
    "gdmcf_tanh_bwd_f32": (c_int, [P, c_int64, P, c_int64, P, c_int64, P, c_int, c_int, P, c_int64, P]),
    "gdmcf_gather_rows_f32": (c_int, [P, c_int64, P, c_int, c_int, P, c_int64, P]),
    "gdmcf_scatter_add_rows_f32": (c_int, [P, c_int64, P, c_int, c_int, P, c_int64, P]),
    "gdmcf_dp_pack_f64": (c_int, [P, P, c_int, P, P, c_int, c_int, c_int, P, P]),
    "gdmcf_dp_unpack_f64": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P]),
    "gdmcf_randn_f32": (c_int, [P, c_int64, c_int, c_int, c_int, c_uint64, c_uint64, P]),
    "gdmcf_eps_target_f32": (c_int, [P, c_int64, P, c_int64, P, c_int64, P, P, P, c_int, c_int, c_int, P, c_int64, P, P, P]),
    "gdmcf_sample_timesteps": (c_int, [P, P, c_int, c_int, c_int, c_double, c_uint64, c_uint64, P, P, P, P]),
    "gdmcf_adamw_f32": (c_int, [P, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_int, c_float, P]),
    "gdmcf_adamw_bf16s_f32": (c_int, [P, P, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_int, c_float, P]),
    "gdmcf_topk_masked_f32": (c_int, [P, c_int64, c_int, c_int, P, P, c_int, P, P, P]),
    "gdmcf_spmm_csr_f32": (c_int, [P, P, P, P, c_int, c_int, P, P, c_int, P, P, c_int, P, c_int64, c_int, P, c_int64, P, P, c_int,
                                   c_int64, c_float, c_double, P]),
    "gdmcf_spmm_bundled_f32": (c_int, [P, c_int, P, P, P, P, c_int, P, P, P, P, c_int, P, P, c_int, P, P, c_int64, c_int, c_int, P, c_int64, c_int, P,
                                       c_int64, P, P, c_int, c_int64, c_float, c_double, P]),
    "gdmcf_spmm_stream_f32": (c_int, [P, c_int, P, c_int64, P, c_int, P, P, c_int, c_int, c_int, P, c_int64, c_int, P, c_int64, P, P,
                                      c_int, c_int64, c_float, c_double, P]),
    "gdmcf_graph_guided_step_u8": (c_int, [P, c_int64, P, c_int, c_int, c_float, P, c_int64, P, P, c_int, c_uint64, c_uint64, P,
                                           c_int64, P, P]),
    "gdmcf_debug_spmm_stamps": (c_int, [c_int, P]),
    "gdmcf_graph_state_bytes": (c_int, []),
    "gdmcf_adam_hyper_bytes": (c_int, []),
    "gdmcf_graph_state_init": (c_int, [P, c_uint64, c_uint64, c_int64, c_int64, c_int64, P]),
    "gdmcf_adam_hyper_fill": (c_int, [P, c_int, c_float, c_float, c_float, c_float, c_float, c_int64, c_float]),
    "gdmcf_graph_state_bind": (c_int, [P]),
    "gdmcf_graph_state_tick": (c_int, [P, P]),
    "gdmcf_row_loss_finish_mean_f64": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, P, P, c_int, P, P, P, P, P, P]),
    "gdmcf_dnn_prep_input_csr_f32": (c_int, [P, P, P, P, P, P, c_int, P, c_int64, c_int, P, c_int64, c_float, c_uint64, c_uint64,
                                             P, P, c_int, c_int, c_int, P, c_int64, P, P, c_int64, P]),
    "gdmcf_linear_loss_fwd_bits_f32": (c_int, [P, c_int64, P, c_int64, P, P, c_int64, P, c_int, c_int, c_int, P, c_int64, P,
                                               c_int64, P, P, P]),
    "gdmcf_scale_f32": (c_int, [P, c_int64, c_float, P, P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load():
    """Loads the shared library (building nothing).  Raises ImportError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m gdmcf_amd.build` (hipcc, gfx950). "
            "gdmcf_amd has no CPU / PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> the .so is stale; rebuild
        fn.restype = res
        fn.argtypes = args
    if lib.gdmcf_version() != 1:
        raise ImportError("libgdmcf_hip.so ABI version mismatch; rebuild with `python -m gdmcf_amd.build --force`")
    _lib = lib
    return lib


_EXC = {E_SHAPE: AssertionError, E_ARG: ValueError, E_UNSUPPORTED: NotImplementedError, E_HIP: RuntimeError,
        E_WORKSPACE: RuntimeError}


def check(rc):
    if rc != GDMCF_OK:
        msg = load().gdmcf_last_error().decode(errors="replace")
        raise _EXC.get(rc, RuntimeError)(msg or f"gdmcf error {rc}")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"gdmcf_amd: {what} must live on the MI355X (got device '{t.device}'); "
                           "the HIP path has no CPU fallback")


def shadow_info(data_ptr):
    """(bf16 pointer, rows, cols, ld) of the registered bf16 shadow of a float32 base pointer, or None."""
    p16 = ctypes.c_void_p()
    rows, cols, ld = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    if not load().gdmcf_bf16_shadow_info(data_ptr, ctypes.byref(p16), ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(ld)):
        return None
    return int(p16.value), int(rows.value), int(cols.value), int(ld.value)


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def philox_randn(shape, device, seed, offset, stream_id=4, out=None):
    """[rows, cols] float32 N(0,1) drawn by gdmcf_randn_f32 (Philox4x32-10, Box-Muller) -- the device draw that stands where
    the reference calls th.randn_like (gaussian_diffusion.py:328-331, :210-217).  No CPU path: `device` must be a GPU."""
    import torch
    rows, cols = int(shape[0]), int(shape[1])
    if out is None:
        out = torch.empty(rows, cols, dtype=torch.float32, device=device)
    require_gpu(out, "philox_randn output")
    check(load().gdmcf_randn_f32(out.data_ptr(), out.stride(0), rows, cols, int(stream_id), int(seed) & (2 ** 64 - 1),
                                 int(offset) & (2 ** 64 - 1), stream_ptr()))
    return out


def schedule_tables(kind, noise_scale, noise_min, noise_max, steps, beta_fixed=True):
    """Host float64 tables [13][T] via the C ABI (no GPU needed)."""
    import numpy as np
    out = np.zeros((N_TABLES, steps), dtype=np.float64)
    rc = load().gdmcf_schedule_build(kind, float(noise_scale), float(noise_min), float(noise_max), int(steps),
                                     int(bool(beta_fixed)), out.ctypes.data_as(c_void_p))
    check(rc)
    return out


class Bf16Shadow:
    """A bf16 shadow of a 2-D float32 device tensor (include/gdmcf_hip.h, gdmcf_bf16_shadow_set): zero-padded
    [round_up(rows, 64)][round_up(cols, 64)] bfloat16 buffer.  Registered on construction; `unregister()` /
    `register()` bracket code that writes the float32 tensor with kernels that do not maintain shadows; the
    registration is dropped with the object."""

    def __init__(self, t, sync=True):
        import weakref

        import torch
        require_gpu(t, "bf16 shadow source")
        if t.dim() != 2 or t.dtype != torch.float32 or t.stride(1) != 1:
            raise RuntimeError("gdmcf_amd: a bf16 shadow needs a 2-D float32 tensor with unit column stride")
        self.rows, self.cols = t.shape
        self.ptr = t.data_ptr()
        self.ld = t.stride(0)
        self.buf = torch.zeros((self.rows + 63) // 64 * 64, (self.cols + 63) // 64 * 64, dtype=torch.bfloat16,
                               device=t.device)
        self._lib = load()
        self._fin = weakref.finalize(self, self._lib.gdmcf_bf16_shadow_clear, self.ptr)
        self.register()
        if sync:
            self.sync()

    def register(self):
        check(self._lib.gdmcf_bf16_shadow_set(self.ptr, self.buf.data_ptr(), self.rows, self.cols, self.buf.stride(0)))

    def unregister(self):
        self._lib.gdmcf_bf16_shadow_clear(self.ptr)

    def sync(self):
        """Refresh the shadow from the float32 tensor (needed after anything but a gdmcf kernel wrote it)."""
        check(self._lib.gdmcf_bf16_shadow_sync(self.ptr, self.ld, stream_ptr()))

    def close(self):
        self._fin()
