/* gdmcf_hip.h -- C ABI of libgdmcf_hip.so, the MI355X (gfx950) hot path of GDMCF.
 *
 * The reference (GDMCF/GDMCF, pure Python on PyTorch) has NO FFI / plugin interface
 * (SURVEY.md F1): its boundary is the Python object protocol between main.py and the classes
 * GaussianDiffusion / DNN / LightGCN.  This header is the C boundary a maintainer would bind
 * underneath those classes; every entry point names the reference code it replaces
 * (file:line relative to the reference root).  gdmcf_amd/ (Python, ctypes) is that binding.
 *
 * Conventions
 *   - plain pointers + sizes; all `const float*`/`float*`/`int64_t*` arguments are DEVICE
 *     pointers unless the name ends in `_host`; no torch types.
 *   - matrices are row-major; `ld*` is the row stride in ELEMENTS (may exceed the width).
 *   - every device function enqueues work on `stream` (a hipStream_t passed as void*) and
 *     returns immediately: 0 = ok, negative = error (GDMCF_E_*); the Python layer maps
 *     GDMCF_E_SHAPE to AssertionError, GDMCF_E_ARG to ValueError, GDMCF_E_UNSUPPORTED to
 *     NotImplementedError (the reference's own error conventions) and
 *     GDMCF_E_HIP to RuntimeError.
 *   - no function allocates device memory; scratch is passed in as `ws`/`ws_bytes`
 *     (query the size with the matching *_ws_bytes function) so everything is
 *     hipGraph-capturable.
 */
#ifndef GDMCF_HIP_H
#define GDMCF_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDMCF_OK 0
#define GDMCF_E_SHAPE (-1)       /* shape/assert violation  -> AssertionError      */
#define GDMCF_E_ARG (-2)         /* bad enum / value        -> ValueError          */
#define GDMCF_E_UNSUPPORTED (-3) /* unknown schedule etc.   -> NotImplementedError */
#define GDMCF_E_HIP (-4)         /* HIP runtime error       -> RuntimeError        */
#define GDMCF_E_WORKSPACE (-5)   /* workspace too small     -> RuntimeError        */

/* library / device ------------------------------------------------------------------ */
int gdmcf_version(void);                 /* ABI version, currently 1 */
const char* gdmcf_last_error(void);      /* thread-local message of the last failing call */
int gdmcf_device_info(int* n_cu, int* wave_size, char* arch_host, int arch_len);
/* Which kernel family the calling thread's LAST dense product was dispatched to (introspection for tests and
 * profiles -- "did the hand-written path run?"): 1 LDS-tiled f32 MFMA kernel, 2 register-streaming weight-gradient
 * kernel, 3 the same with the AdamW stream, 4 fat-tile output-layer kernel (one tile per wave), 5 hybrid kernel
 * (opt-in), 6 register-streaming forward kernel (opt-in), 7 bf16-input kernels, 8 f32x3 kernels, 9 element-wise
 * kernel for degenerate shapes; 0 none yet.                                                              */
int gdmcf_debug_last_gemm(void);

/* Optional in-library timing with HIP events on the launch stream (bench.py's live roofline
 * measurement).  While enabled, every tagged kernel launch is bracketed by two hipEventRecord
 * calls; gdmcf_prof_collect synchronises them and returns (tag, milliseconds, work) triples,
 * where work = algorithmic FLOPs (GEMM tags) or bytes (HBM-bound tags) of that launch.
 * Tags: 1 linear_fwd gemm, 2 loss_fwd gemm, 3 posterior gemm, 4 bwd_input gemm, 5 bwd_weight gemm,
 * 6 adamw, 7 prep_input, 8 spmm, 9 topk.
 * on: 1 = record, 2 = pause (stop recording, keep the records), 0 = off and discard.           */
int gdmcf_prof_enable(int on);
int gdmcf_prof_collect(int cap, int* tags_host, float* ms_host, double* work_host);

/* ---- schedules: host, float64 --------------------------------------------------------
 * replaces GaussianDiffusion.get_betas + calculate_for_diffusion
 * (models/gaussian_diffusion.py:109-159, betas_from_linear_variance :1138-1144,
 * betas_for_alpha_bar :1146-1163, betas[0]=1e-5 :79).
 * kind: 0 linear, 1 linear-var, 2 cosine, 3 binomial.
 * out_tables_host: [13][T] doubles in this order: betas, alphas_cumprod, alphas_cumprod_prev,
 * alphas_cumprod_next, sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod,
 * log_one_minus_alphas_cumprod, sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod,
 * posterior_variance, posterior_log_variance_clipped, posterior_mean_coef1, posterior_mean_coef2.
 * Returns GDMCF_E_SHAPE when the reference's asserts (:81-83) would fire. */
#define GDMCF_N_TABLES 13
int gdmcf_schedule_build(int kind, double noise_scale, double noise_min, double noise_max, int T,
                         int beta_fixed, double* out_tables_host);

/* ---- batch provider: dense user rows from a device-resident CSR interaction matrix ----------
 * replaces DataDiffusion.__getitem__ + DataLoader collation + batch.to(device)
 * (data_utils.py:216-226, main.py:155,343): out[b, 0:I] = dense row row_ids[b] (row_ids NULL -> b).
 * indptr int64 [n_users+1], indices int32, values float32 or NULL (all ones); rows must hold each
 * column at most once (scipy's csr_matrix constructor has already summed duplicates).        */
int gdmcf_densify_rows_f32(const int64_t* indptr, const int32_t* indices, const float* values,
                           const int64_t* row_ids, int B, int I, float* out, int64_t ldo, void* stream);

/* ---- denoiser input: q_sample + F.normalize + dropout + timestep embedding + cat ------
 * replaces GaussianDiffusion.q_sample (:399-407) with _extract_into_tensor (:532-547),
 * and DNN.forward's prologue (models/DNN.py:73-78): timestep_embedding (:1806-1825),
 * emb_layer, F.normalize, nn.Dropout, torch.cat.
 *   xin[b, 0:I]   = drop( norm( ca[ts[b]]*x[b,:] + cb[ts[b]]*noise[b,:] ) )
 *   xin[b, I:I+E] = emb_w @ temb(ts[b]) + emb_b ;  xin[b, I+E:ldxin] = 0
 * ca/cb: float32 tables [T] (the reference casts the f64 tables to f32 before use); pass
 * NULL for both to skip the q_sample (xin <- x).
 * noise_mode 0: none (requires ca==NULL or cb treated as 0); 1: explicit `noise` [B,ldn];
 *            2: Philox4x32-10 + Box-Muller N(0,1): element (b,i) is component i&3 of the block with
 *               counter (i>>2, b, stream, offset) and key seed (stream 0 = noise, 1 = dropout),
 *               so any kernel can regenerate it.
 * drop_mode  0: none; 1: explicit keep-mask `keep` (uint8 [B,ldkeep]); 2: Philox Bernoulli(1-p): element (b,i) is kept iff its
 *               16-bit uniform < round((1-p) * 65536) (the keep probability is quantised to 2^-16: exact for p = 0.5; the scale
 *               1/(1-p) is not quantised); the uniform is half (i >> 10) & 1 (low, high) of word i & 3 of the stream-1 block with
 *               counter ((i & ~1024) >> 2, b, 1, offset): one block serves the eight elements i..i+3 and i+1024..i+1027.
 * normalize != 0 applies F.normalize (L2, eps 1e-12) to the (noised) row before dropout and
 * needs rownorm_ws (float32 [B] scratch).
 * xt_out (optional, [B,ldxt]) receives the pre-normalize/pre-dropout x_t.
 * temb_out (optional, [B,E]) receives the sinusoidal embedding (needed by the backward).   */
int gdmcf_dnn_prep_input_f32(const float* x, int64_t ldx, const int64_t* ts, const float* ca,
                             const float* cb, int noise_mode, const float* noise, int64_t ldn,
                             int drop_mode, const uint8_t* keep, int64_t ldkeep, float drop_p,
                             uint64_t seed, uint64_t offset, int normalize, const float* emb_w,
                             const float* emb_b, int E, int B, int I, float* xin, int64_t ldxin,
                             float* xt_out, int64_t ldxt, float* temb_out, float* rownorm_ws,
                             void* stream);
/* The same builder fed from a device-resident CSR matrix of {0,1} interactions instead of dense
 * rows: row b of the batch is row rows[b] of (indptr int64, indices int32).  Nothing dense is read: a workgroup marks its
 * 4096 columns of the row in an LDS bitmap and builds x_t = ca*x0 + cb*noise, dropout, the embedding columns from it
 * (same arithmetic and the same Philox streams as the dense entry: identical xin).  bits_out (optional, uint32
 * [B, ldbits], ldbits >= ceil(I/32)) receives the rows as bitmaps -- the target of gdmcf_linear_loss_fwd_bits_f32.
 * No F.normalize here (it needs the dense row twice): densify (gdmcf_densify_rows_f32) and use the dense entry.        */
int gdmcf_dnn_prep_input_csr_f32(const int64_t* indptr, const int32_t* indices, const int64_t* rows, const int64_t* ts,
                                 const float* ca, const float* cb, int noise_mode, const float* noise, int64_t ldn,
                                 int drop_mode, const uint8_t* keep, int64_t ldkeep, float drop_p, uint64_t seed,
                                 uint64_t offset, const float* emb_w, const float* emb_b, int E, int B, int I, float* xin,
                                 int64_t ldxin, float* temb_out, uint32_t* bits_out, int64_t ldbits, void* stream);

/* ---- one-hot rows with discrete transition noise (first slice) -------------------
 * replaces GaussianDiffusionDiscrete.apply_noise (gaussian_diffusion.py:770-831: get_Qt_bar :597-614,
 * sample_discrete_features :999-1038) together with F.one_hot(x_start) and the `x_tU & one_hot(x_start)` that
 * follows it (:841-849 in training_losses, :672-686 in p_sample), and the x_U.reshape of DNNOneHot.forward
 * (models/DNN.py:444):
 *   c0 = x0[b,i] != 0 ; s ~ row c0 of Q = a*I + (1-a)*[[e,1-e],[e,1-e]], a = (float)ts[b]/B (:775), e = discrete ;
 *   xU[b, 2i + c] = (c == c0 && s == c0) ? 1 : 0          (float32, the [B, 2I] input of the second MLP branch)
 * sampled (optional, uint8 [B,lds]): the drawn classes s are given (parity runs; ts may be NULL then); otherwise
 * s = (u < P(1)) with u the Philox4x32-10 uniform of element (b,i): component i&3 of the block with counter
 * (i>>2, b, 3, offset), key seed.  sampled_out (optional) receives s.                                        */
int gdmcf_onehot_noise_f32(const float* x0, int64_t ldx, const int64_t* ts, int B, int I, float discrete,
                           const uint8_t* sampled, int64_t lds, uint64_t seed, uint64_t offset,
                           float* xU, int64_t ldu, uint8_t* sampled_out, int64_t ldso, void* stream);

/* ---- N(0,1) fill --------------------------------------------------------------------------------
 * replaces `noise = th.randn_like(x_start)` where the noise itself is needed in memory: the eps TARGET of
 * training_losses (gaussian_diffusion.py:328-331, :844-846) and the reverse loop's `noise = th.randn_like(x_t)`
 * (:210-217, :696-703).  out[b, i] (float32 [rows, ld]) = normal (i & 3) of the Philox4x32-10 block with counter
 * (i >> 2, b, stream_id, (uint32)offset), key seed: Box-Muller on the block's (x, y) and (z, w) words, hardware log / sqrt /
 * sin / cos.  stream_id 0 is the stream gdmcf_dnn_prep_input(_csr)_f32 draws in place (noise_mode 2): a buffer filled here
 * with the same (seed, offset) and handed to it as given noise (noise_mode 1) yields the same x_t bit for bit.  Stream ids
 * 1-3, 5, 6 belong to dropout / timestep / one-hot / graph draws; 4 and 7 are free for callers (eps target, step noise).  */
int gdmcf_randn_f32(float* out, int64_t ld, int rows, int cols, int stream_id, uint64_t seed, uint64_t offset, void* stream);

/* ---- loss target of the eps parameterisation ---------------------------------------------------
 * replaces the element-wise passes of training_losses for ModelMeanType.EPSILON (gaussian_diffusion.py:328-348):
 *   target[b,:] = noise[b,:]                      alpha[b] = 1       rowdiv[b] = I      (t != 0, or t0_likelihood == 0)
 *   target[b,:] = r1[0]*x_t[b,:] - x0[b,:]         alpha[b] = r2[0]   rowdiv[b] = 2 I    (t == 0: the x0-likelihood row, :344-348)
 * (product and difference rounded separately, as torch's mul and sub).  alpha / rowdiv feed gdmcf_linear_loss_fwd_f32 and
 * gdmcf_row_loss_finish_*.  target may equal noise (same leading dimension): only the t == 0 rows are written then.      */
int gdmcf_eps_target_f32(const float* noise, int64_t ldn, const float* xt, int64_t ldxt, const float* x0, int64_t ldx0,
                         const int64_t* ts, const float* r1, const float* r2, int t0_likelihood, int B, int I, float* target,
                         int64_t ldt, float* alpha, float* rowdiv, void* stream);

/* ---- degree-guided graph of the reverse loop (gaussian_diffusion.py:706-729, inside GaussianDiffusionDiscrete.p_sample) ----
 * One reverse step's update of the accumulated user-item graph, one byte per edge state:
 *   s[b,i] ~ row c of Q = a*I + (1-a)*[[e,1-e],[e,1-e]], c = graph[b,i], a = (float)ts[b]/B, e = discrete   (apply_noise on
 *            the accumulated one-hot graph, :709);
 *   pick[b] ~ Bernoulli(degree_prob[b]), degree_prob = row sum / largest row sum of x_start   (multinomial(1), :710-716);
 *   graph[b,i] |= s[b,i] & (user_guided ? pick[b] : 1)                                          (:719-727).
 * sampled_in / pick_in (optional, uint8): the draws are given (parity runs); otherwise Philox4x32-10, key seed, counters
 * (i>>2, b, 5, offset) for the classes and (0xFFFFFFFF, b, 6, offset) for the user bit.  sampled_out / pick_out
 * (optional) receive the draws.  The graph is what the reference hands to the denoiser as `graph=` (:744).              */
int gdmcf_graph_guided_step_u8(uint8_t* graph, int64_t ldg, const int64_t* ts, int B, int I, float discrete,
                               const uint8_t* sampled_in, int64_t lds, const uint8_t* pick_in, const float* degree_prob,
                               int user_guided, uint64_t seed, uint64_t offset, uint8_t* sampled_out, int64_t ldso,
                               uint8_t* pick_out, void* stream);

/* Rewrites only the embedding + padding columns [I, ldxin) of xin for new timesteps (reverse
 * loop: x_t already sits in xin[:, 0:I], written by gdmcf_linear_posterior_fwd_f32).          */
int gdmcf_dnn_emb_cols_f32(const int64_t* ts, const float* emb_w, const float* emb_b, int E, int B,
                           int I, float* xin, int64_t ldxin, float* temb_out, void* stream);

/* ---- dense layers on the f32 MFMA (v_mfma_f32_16x16x4_f32) -------------------------------
 * replaces nn.Linear (+tanh) in DNN.forward (models/DNN.py:79-86) and their autograd backward
 * (main.py:350).  W is nn.Linear layout [N,K].  act: 0 none, 1 tanh.                        */
/* Input precision of the dense products (BASELINE configs[2] "bf16 denoiser GEMM on MFMA").  The setting is
 * per calling thread and applies to every gdmcf_linear_* call that follows; returns the previous mode (a mode
 * outside the enum only queries).  GDMCF_GEMM_BF16: both operands of each product are rounded to bfloat16
 * (nearest-even) on chip and multiplied on the bf16 matrix pipe with float32 accumulation; all tensors in HBM
 * (weights, activations, gradients, optimiser state) stay float32.  Replaces what the reference would obtain
 * with torch.autocast(dtype=torch.bfloat16) around models/DNN.py:79-86 -- the reference itself runs fp32.
 * GDMCF_GEMM_F32X3: float32 products on the bf16 matrix pipe -- every operand is split on chip into three bfloat16
 * terms (a = a0 + a1 + a2, exact to 2^-26 |a|) and each product is assembled from the six partial products above
 * 2^-25 |a b| with float32 accumulation: float32-level error (no operand is rounded), 2.67x the matrix-pipe rate of
 * v_mfma_f32_16x16x4_f32; non-finite inputs give NaN.  No shadows, nothing extra in HBM.                     */
enum { GDMCF_GEMM_F32 = 0, GDMCF_GEMM_BF16 = 1, GDMCF_GEMM_F32X3 = 2 };
int gdmcf_gemm_precision(int mode);
/* bf16 shadows (GDMCF_GEMM_BF16 only): a shadow is a bfloat16 copy of a float32 matrix that the library may
 * stream INSTEAD of the float32 matrix when that matrix is an operand of a dense product (half the bytes, no
 * conversion on chip), and that these library kernels keep up to date when they write the float32 matrix:
 * gdmcf_dnn_prep_input_f32 / gdmcf_dnn_emb_cols_f32 (xin), gdmcf_linear_fwd_f32 (C), gdmcf_linear_loss_fwd_f32
 * (diff), gdmcf_linear_posterior_fwd_f32 (x_next), gdmcf_linear_bwd_input_f32 (dA), gdmcf_rowscale_f32 (out),
 * gdmcf_adamw_bf16s_f32 and gdmcf_linear_bwd_weight_adamw_f32 (W).  The float32 matrix stays authoritative; a
 * shadow whose float32 matrix was written by anything else must be refreshed with gdmcf_bf16_shadow_sync.  Layout: [round_up(rows, 64)][ld_bf16] bfloat16, ld_bf16 a multiple of 64
 * and >= cols, 16-byte aligned, ZERO outside [rows, cols) (the kernels rely on the zero padding instead of edge
 * predicates).  The registry is keyed by the float32 base pointer and global to the process: clear an entry
 * before its buffers are freed.  A product uses shadows only when BOTH operands have one of exactly its shape. */
int gdmcf_bf16_shadow_set(const float* f32, void* bf16, int64_t rows, int64_t cols, int64_t ld_bf16);
int gdmcf_bf16_shadow_clear(const float* f32 /* NULL: all */);
void* gdmcf_bf16_shadow_get(const float* f32);
/* 1 and the registered description, or 0 when `f32` has no shadow */
int gdmcf_bf16_shadow_info(const float* f32, void** bf16, int64_t* rows, int64_t* cols, int64_t* ld_bf16);
int gdmcf_bf16_shadow_sync(const float* f32, int64_t ld, void* stream);
size_t gdmcf_linear_ws_bytes(int M, int N, int K);
/* C[M,N] = act(A[M,K] @ W[N,K]^T + bias) */
int gdmcf_linear_fwd_f32(const float* A, int64_t lda, const float* W, int64_t ldw,
                         const float* bias, int act, int M, int N, int K, float* C, int64_t ldc,
                         void* ws, size_t ws_bytes, void* stream);
/* The same layer with the weight given TRANSPOSED: Wt[K, N] row-major (ldwt >= N), out = act(A @ Wt + bias).  For loops over
 * FROZEN weights (the reverse-diffusion loop of evaluation, gaussian_diffusion.py:161-220 over models/DNN.py:79-81): with a
 * K-contiguous A and an N-contiguous Wt the product runs on the register-streaming kernel of the input gradient (both operands
 * straight into the MFMA layout, no LDS).  The caller keeps Wt equal to W^T; float32 GEMM mode only; ws as gdmcf_linear_ws_bytes. */
int gdmcf_linear_fwd_wt_f32(const float* A, int64_t lda, const float* Wt, int64_t ldwt,
                            const float* bias, int act, int M, int N, int K, float* C, int64_t ldc,
                            void* ws, size_t ws_bytes, void* stream);
/* Last layer fused with the per-row diffusion loss (gaussian_diffusion.py:335 mean_flat):
 *   out = A @ W^T + bias ;  diff[m,n] = alpha[m]*out[m,n] - target[m,n]   (alpha NULL -> 1)
 *   rowsum[m] = sum_n diff[m,n]^2  (deterministic two-stage reduction)
 * `out` may be NULL (training never needs it).  rowpart: scratch [M, gdmcf_loss_tiles(N)]. */
int gdmcf_loss_tiles(int N);
int gdmcf_linear_loss_fwd_f32(const float* A, int64_t lda, const float* W, int64_t ldw,
                              const float* bias, const float* target, int64_t ldt,
                              const float* alpha, int M, int N, int K, float* out, int64_t ldo,
                              float* diff, int64_t ldd, float* rowpart, float* rowsum,
                              void* stream);
/* The same product with the target given as BITMAPS of {0,1} rows (word n>>5 of row m, bit n&31; row stride ldbits words)
 * instead of a dense float matrix: the CSR input path (gdmcf_dnn_prep_input_csr_f32 writes the bitmaps) never densifies
 * the batch.  x0-target training only (the eps target is the noise); results are bit-identical to the dense entry.    */
int gdmcf_linear_loss_fwd_bits_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                                   const uint32_t* target_bits, int64_t ldbits, const float* alpha, int M, int N, int K,
                                   float* out, int64_t ldo, float* diff, int64_t ldd, float* rowpart, float* rowsum,
                                   void* stream);
/* Last layer fused with the reverse-diffusion posterior mean (gaussian_diffusion.py:473-515,
 * :451-471, :518-523, :210-217):  out = A @ W^T + bias
 *   pred = eps_mode ? r1[t]*x_t - r2[t]*out : out ;  mean = c1[t]*pred + c2[t]*x_t
 *   x_next = mean + (t!=0 ? sigma[t]*z : 0)            (z NULL -> deterministic)
 * per-row coefficient vectors c1,c2,r1,r2,sigma are float32 [M] (already gathered by t).    */
int gdmcf_linear_posterior_fwd_f32(const float* A, int64_t lda, const float* W, int64_t ldw,
                                   const float* bias, const float* x_t, int64_t ldxt,
                                   const float* c1, const float* c2, const float* r1,
                                   const float* r2, const float* sigma, const float* z,
                                   int64_t ldz, int M, int N, int K, float* x_next,
                                   int64_t ldxn, float* pred_out, int64_t ldp, void* stream);
/* dA[M,K] = rowscale[m] * (dZ[M,N] @ W[N,K]) * (act==1 ? 1 - Aact[m,k]^2 : 1)
 * (grad wrt the layer input, fused with the previous layer's tanh').  rowscale may be NULL. */
int gdmcf_linear_bwd_input_f32(const float* dZ, int64_t lddz, const float* W, int64_t ldw,
                               const float* rowscale, const float* Aact, int64_t ldact, int act,
                               int M, int N, int K, float* dA, int64_t ldda, void* ws,
                               size_t ws_bytes, void* stream);
/* dW[N,K] = dZ[M,N]^T @ A[M,K]  (weight grad; accumulate != 0 adds into dW)
 * db[N]   = sum_m rowscale[m]*dZ[m,n]   (db NULL -> skipped).  When rowscale != NULL the
 * caller passes A already multiplied by rowscale (see gdmcf_rowscale_f32).
 * a_scale_col != 0 is the caller's statement that lda > K and A[m, K] == rowscale[m] for every
 * row m (== 1 when rowscale is NULL) -- which is how gdmcf_rowscale_f32 leaves its output when
 * ldo > K, and how gdmcf_dnn_prep_input_*_f32 leave xin (column I + E holds 1 when ldxin > I + E):
 * db then comes out of the product as its column K (no second pass over dZ).  The library keeps NO record between
 * calls and does not check the column; with a_scale_col == 0 (any A, any column K) db takes a
 * column-sum pass over dZ -- the same result within float32 rounding.  Kernels that cannot
 * carry the extra column (bf16 / f32x3 modes, small shapes) ignore the flag and take the pass. */
int gdmcf_linear_bwd_weight_f32(const float* dZ, int64_t lddz, const float* A, int64_t lda,
                                const float* rowscale, int a_scale_col, int M, int N, int K,
                                float* dW, int64_t lddw, float* db, int accumulate, void* stream);
/* Weight gradient fused with the AdamW update of that weight (single-GPU optimiser-in-backward):
 * G = dZ^T @ A stays in the MFMA accumulators and W, exp_avg, exp_avg_sq ([N,K], row stride ldw, all three) are
 * updated in the epilogue with torch.optim.AdamW's single-tensor math for step number `step`
 * (32 -> 24 B/param of HBM traffic; the gradient is never materialised).  W must not be read by later
 * kernels of the same backward pass (the caller computes the input gradient first).  a_scale_col: as above. */
int gdmcf_linear_bwd_weight_adamw_f32(const float* dZ, int64_t lddz, const float* A, int64_t lda,
                                      const float* rowscale, int a_scale_col, int M, int N, int K, float* W,
                                      int64_t ldw, float* exp_avg, float* exp_avg_sq, float* db, float lr,
                                      float beta1, float beta2, float eps, float weight_decay, int step,
                                      float grad_scale, void* stream);
/* out[m, k] = rowscale[m] * A[m, k], k < K (the scaled activation copy of the weight-gradient product: (rs . dZ)^T A ==
 * dZ^T (rs . A)); with ldo > K also out[m, K] = rowscale[m]: the column gdmcf_linear_bwd_weight_*'s a_scale_col speaks of. */
int gdmcf_rowscale_f32(const float* A, int64_t lda, const float* rowscale, int M, int K, float* out,
                       int64_t ldo, void* stream);
/* ---- pieces of the indexIn backbone DNNOneHotEmbedding (models/DNN.py:510-682) ----------------
 * Its output layer is a cosine similarity (:655, :667-682): scores = (u @ V^T) / (|u| |v|^T) with u = [h, h_U,
 * embedding_user(index)] and V = embedding_item.weight.  The products run on gdmcf_linear_* with the row-normalised
 * operands (gdmcf_rowscale_f32 by the inverse norms); these entries are the HBM-bound glue around them:
 *   row_norms:          norm[r] = |X[r,:]|, inv_norm[r] = 1/|X[r,:]|                     (torch.norm(dim=1), :675-676)
 *   normalize_rows_bwd: dX = (dY - Y * <dY, Y>) * inv_norm for Y = X/|X| (backward of the division; dX may alias dY)
 *   tanh_bwd:           out = (dA + scale[0] * extra) * (1 - A^2)   (tanh' of the two hidden activations whose
 *                       gradient also receives the NT-Xent term's, DNN.py:641-643; extra NULL -> no addend;
 *                       scale is a device scalar)
 *   gather_rows:        dst[j,:] = src[index[j],:]                                        (nn.Embedding lookup, :650)
 *   scatter_add_rows:   dst[index[j],:] += src[j,:]                                       (its backward)            */
int gdmcf_row_norms_f32(const float* X, int64_t ld, int rows, int cols, float* norm, float* inv_norm,
                        void* stream);
int gdmcf_normalize_rows_bwd_f32(const float* dY, int64_t lddy, const float* Y, int64_t ldy,
                                 const float* inv_norm, int rows, int cols, float* dX, int64_t lddx,
                                 void* stream);
int gdmcf_tanh_bwd_f32(const float* dA, int64_t ldd, const float* A, int64_t lda, const float* extra,
                       int64_t lde, const float* scale, int M, int N, float* out, int64_t ldo,
                       void* stream);
int gdmcf_gather_rows_f32(const float* src, int64_t lds, const int64_t* index, int n, int cols, float* dst,
                          int64_t ldd, void* stream);
int gdmcf_scatter_add_rows_f32(const float* src, int64_t lds, const int64_t* index, int n, int cols,
                               float* dst, int64_t ldd, void* stream);
/* Gradients of the timestep-embedding branch (models/DNN.py:73-74,78):
 *   demb[m,e] = sum_n dZ1[m,n]*W1[n, I+e] ;  dWe = demb^T @ temb ;  dbe = sum_m demb
 * demb_ws: float32 scratch of (M + N) * E elements.                                         */
int gdmcf_emb_bwd_f32(const float* dZ1, int64_t lddz, const float* W1, int64_t ldw, int I, int E,
                      const float* temb, int M, int N, float* demb_ws, float* dWe, float* dbe,
                      void* stream);

/* ---- per-row loss tail, float64 (gaussian_diffusion.py:339-370) --------------------------
 *   loss[b] = weight_t[ts[b]] * (float64)(rowsum[b]/rowdiv[b]) ; Lt-history FIFO update in
 *   batch order (:355-368) ; loss[b] /= pt[b].   rowdiv = I (mse) or 2I (likelihood), f32.
 * weight_t: float64 [T].  pt: float64 [B].  Lt_history: float64 [T,H]; Lt_count: int64 [T].
 * gradcoef[b] (optional) = d loss[b] / d out[b,n] / diff[b,n] = 2*alpha[b]*weight/(pt*rowdiv),
 * the per-row factor the backward multiplies `diff` by (alpha NULL -> 1).
 * update_history == 0 leaves the ring buffer untouched (data-parallel ranks call
 * gdmcf_lt_history_update on the gathered batch instead).                                  */
int gdmcf_row_loss_finish_f64(const float* rowsum, const float* rowdiv, const float* alpha,
                              const int64_t* ts, const double* weight_t, const double* pt, int B,
                              int T, int H, double* Lt_history, int64_t* Lt_count,
                              int update_history, double* loss_unscaled, double* loss,
                              float* gradcoef, void* stream);
/* The same tail, additionally emitting what the reference's step computes next (main.py:348-350): loss_mean (optional,
 * float64 scalar) = mean_b loss[b], summed in a fixed order, and rowscale_mean[b] (optional, float32) = gradcoef[b] *
 * (float)(1/B) -- the per-row scale of the backward of that mean -- so that the step needs no reduction / scaling launch. */
int gdmcf_row_loss_finish_mean_f64(const float* rowsum, const float* rowdiv, const float* alpha, const int64_t* ts,
                                   const double* weight_t, const double* pt, int B, int T, int H, double* Lt_history,
                                   int64_t* Lt_count, int update_history, double* loss_unscaled, double* loss,
                                   float* gradcoef, double* loss_mean, float* rowscale_mean, void* stream);
int gdmcf_lt_history_update(const int64_t* ts, const double* loss_unscaled, int B, int T, int H,
                            double* Lt_history, int64_t* Lt_count, void* stream);

/* ---- data-parallel exchange helpers (new: the reference is single-process; main.py:345-351 is the step whose
 * gradients are exchanged).  One float64 buffer `flat` carries, per rank and step,
 *   [ sum(counts) ]   the n (<= 16) small float32 gradients, concatenated in table order, and
 *   [world][B][2]     (ts, unscaled loss) of every rank's rows -- a rank writes its own slice, zeros elsewhere,
 * so a single SUM all-reduce of `flat` both reduces the small gradients and gathers the history inputs in rank
 * order (what gdmcf_lt_history_update then replays on the global batch).  `grads` / `counts` are HOST arrays of
 * device pointers / element counts (copied into the kernel arguments).
 * pack: flat <- this rank's contribution.  unpack: gradients <- (float)flat, ts_all[world*B] (int64) and
 * loss_unscaled_all[world*B] (float64) contiguous in rank order.                                              */
int gdmcf_dp_pack_f64(const float* const* grads, const int64_t* counts, int n, const int64_t* ts,
                      const double* loss_unscaled, int B, int rank, int world, double* flat,
                      void* stream);
int gdmcf_dp_unpack_f64(const double* flat, float* const* grads, const int64_t* counts, int n, int B,
                        int world, int64_t* ts_all, double* loss_unscaled_all, void* stream);

/* ---- importance-sampled timesteps (gaussian_diffusion.py:373-397), one launch, no host sync ----
 * Until every Lt_count == H: t ~ uniform{0..T-1}, pt = 1.  Afterwards p = sqrt(mean(Lt_history^2))
 * normalised, mixed as p*(1-uniform_prob) + uniform_prob/T; t by inverse CDF, pt = p[t]*T.
 * Random numbers: Philox4x32-10 (counter (b,0,2,offset), key seed) -- the reference's torch.randint /
 * torch.multinomial streams cannot be reproduced on the device, parity tests inject ts instead.
 * p_out (optional, float64 [T]) receives the probability vector when the importance branch is live. */
int gdmcf_sample_timesteps(const double* Lt_history, const int64_t* Lt_count, int T, int H, int B,
                           double uniform_prob, uint64_t seed, uint64_t offset, int64_t* ts,
                           double* pt, double* p_out, void* stream);

/* ---- fused multi-tensor AdamW (torch.optim.AdamW as used at main.py:258,351) -------------
 * table: device int64 [n_tensors][6] = {param*, grad*, exp_avg*, exp_avg_sq*, numel, first_block}
 * (first_block = prefix sum of ceil(numel/4096)); total_blocks = grid size.
 * grad_scale multiplies every gradient first (1/world_size for data parallel).             */
int gdmcf_adamw_f32(const int64_t* table, int n_tensors, int total_blocks, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale,
                    void* stream);
/* Same update, additionally storing each updated parameter rounded to bfloat16 into its registered shadow:
 * shadow_table [n_tensors][3] int64 device array = (shadow pointer or 0, columns of the 2-D parameter, shadow row
 * stride) as returned by gdmcf_bf16_shadow_info.  Parameters must have fewer than 2^32 elements.              */
int gdmcf_adamw_bf16s_f32(const int64_t* table, const int64_t* shadow_table, int n_tensors, int total_blocks, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                          void* stream);

/* ---- evaluation: history mask + top-k (main.py:296-301) ----------------------------------
 * For every row: entries listed in the CSR history mask become -inf, then the k largest
 * scores are returned in descending order; ties broken by LOWEST index (torch.topk leaves
 * tie order unspecified).  mask_indptr int64 [B+1] / mask_indices int32 (NULL = no mask).  */
int gdmcf_topk_masked_f32(const float* pred, int64_t ldp, int B, int I, const int64_t* mask_indptr,
                          const int32_t* mask_indices, int k, int64_t* idx_out, float* val_out,
                          void* stream);

/* Ranking metrics of reference evaluate_utils.py:6-52 (computeTopNAccuracy) on the device: per user and per cut-off
 * N of topN_host (ascending, at most 8) the four terms precision = hits/N, recall = hits/|GT|, NDCG = dcg/idcg,
 * MRR = 1/(rank of the first hit), all 0 for a user with an empty ground truth -- float64, accumulated in rank order
 * exactly as the reference's loop, written to out[U][n_topn][4].  pred_idx: [U, >= max N] item ids (row stride ldp);
 * ground truth as CSR with sorted column indices.  The caller adds the terms up over users and divides by U.        */
int gdmcf_topn_metrics_f64(const int64_t* pred_idx, int64_t ldp, int U, const int64_t* gt_indptr, const int32_t* gt_indices,
                           const int* topN_host, int n_topn, double* out, void* stream);

/* ---- LightGCN propagation: CSR SpMM (lightGCN.py:184-189) --------------------------------
 * Y[r,:] = ( sum_j val[j]*X[col[j],:]  +  sum_k addend_k[r,:] ) * scale
 * The adjacency is plain CSR (col int32, val float32) plus a host-built execution plan of "virtual rows"
 * (gdmcf_amd/lightgcn.py:spmm_plan):
 *   vbeg, vend int64 [n_virtual]  nonzero range of each virtual row
 *   vrow  int32 [n_virtual]       the real row it belongs to
 *   vslot int32 [n_virtual]       -1: the row is whole, write Y directly; >= 0: slot in partial_ws
 *   the first n_short virtual rows are WHOLE rows of at most a few dozen nonzeros: they run 64/(d/4) rows per
 *   wave (four at d = 64), because one-row-per-wave is latency bound on such rows;
 *   the remaining ones are pieces of <= chunk nonzeros, one wave each, so hub nodes spread over many waves;
 *   lrow  int32 [n_long], lptr int32 [n_long+1]: the rows that were cut and their slot ranges in
 *   partial_ws float32 [n_slots, d], added up in slot order by a second small kernel (no atomics).
 * With n_short = 0 and chunk = infinity the plan degenerates to plain CSR (vbeg = rowptr[:-1], vend = rowptr[1:]).
 * addends_host: HOST array of n_add (<= 8) device pointers [n_rows, ld_add] -- the last LightGCN
 * layer passes E_0..E_{L-1} and scale = 1/(L+1) so the layer mean (:188-189) costs no extra pass.
 * alg_bytes: algorithmic bytes of this launch, forwarded to the profiling hook only.          */
int gdmcf_spmm_csr_f32(const int64_t* vbeg, const int64_t* vend, const int32_t* vrow, const int32_t* vslot,
                       int n_virtual, int n_short, const int32_t* lrow, const int32_t* lptr, int n_long, const int32_t* col,
                       const float* val, int n_rows, const float* X, int64_t ldx, int d, float* Y,
                       int64_t ldy, float* partial_ws, const float* const* addends_host, int n_add,
                       int64_t ld_add, float scale, double alg_bytes, void* stream);
/* Second-generation schedule of the same product (one launch per layer; csrc/spmm_bundle.hip), for d in
 * {8,16,32,64,128,256} with 16-byte aligned rows.  The host plan (gdmcf_amd/lightgcn.py:spmm_bundle_plan) sorts the rows
 * of at most s_max nonzeros by length and bundles them G = 64/(d/4) to a wave-step, cuts longer rows into pieces, and
 * gives every one of n_waves waves (a multiple of 32; block b = 4 waves serves class b % 8 = one XCD) a contiguous run
 * of pieces and bundles of equal cost:
 *   wdesc int32 [n_waves][4]   first / last+1 piece and first / last+1 bundle of the wave (class-major order)
 *   lbeg int64, llen/lrow/lslot int32 [n_pieces]   first nonzero, length, row, partial slot (-1: the row is whole)
 *   sbeg int64, slen/srow int32 [n_bundles*G], smax int32 [n_bundles]   per bundle entry: first nonzero, length, row
 *                                                  (-1 = padding), and the longest length of the bundle
 *   crow int32 [n_cut], cptr int32 [n_cut+1]       rows cut into several pieces and their slot ranges in partial_ws
 * nnz = length of col / val (> 0), n_x_rows = rows of X (columns of the matrix).  Everything else as
 * gdmcf_spmm_csr_f32.  Returns GDMCF_E_UNSUPPORTED for other widths / alignments.               */
int gdmcf_spmm_bundled_f32(const int32_t* wdesc, int n_waves, const int64_t* lbeg, const int32_t* llen, const int32_t* lrow,
                           const int32_t* lslot, int n_pieces, const int64_t* sbeg, const int32_t* slen, const int32_t* srow,
                           const int32_t* smax, int n_bundles, const int32_t* crow, const int32_t* cptr, int n_cut,
                           const int32_t* col, const float* val, int64_t nnz, int n_rows, int n_x_rows, const float* X,
                           int64_t ldx, int d, float* Y, int64_t ldy, float* partial_ws, const float* const* addends_host, int n_add, int64_t ld_add,
                           float scale, double alg_bytes, void* stream);
/* Third generation: the schedule of gdmcf_spmm_bundled_f32 with the nonzeros re-ordered into the order the waves gather
 * them (gdmcf_amd/lightgcn.py:spmm_stream_pack), so that a wave reads one contiguous run of (col, val) pairs instead of
 * chasing row pointers:
 *   wdesc int32 [n_waves][4]     first 64-entry batch of the wave's run, its batches, first / last+1 unit
 *   cw    int32 [n_entries][2]   (column, float bits of the value); n_entries % 64 == 0; 8-byte aligned
 *   ud    int32 [n_units][DW]    DW = 1 + max(G, 2), G = 64/(d/4):  [steps/UN | piece << 31,  piece: row, slot (-1 = whole
 *                                row) | bundle: row of lane group g (-1 = padding, bit 30 = empty row)], UN = min(4, d/4)
 *   crow / cptr / partial_ws / addends / scale / alg_bytes as above.                                                     */
int gdmcf_spmm_stream_f32(const int32_t* wdesc, int n_waves, const int32_t* cw, int64_t n_entries, const int32_t* ud, int n_units,
                          const int32_t* crow, const int32_t* cptr, int n_cut, int n_rows, int n_x_rows, const float* X,
                          int64_t ldx, int d, float* Y, int64_t ldy, float* partial_ws, const float* const* addends_host,
                          int n_add, int64_t ld_add, float scale, double alg_bytes, void* stream);
/* development aid (tools/spmm_waves.py): per-wave timestamps of gdmcf_spmm_stream_f32's main kernel.  (n_waves, NULL) starts
 * recording 8 int64 per wave -- start, first gather, end of pieces, end (100 MHz ticks), units, batches, XCC id, block --,
 * (n_waves, host buffer) copies them out and stops.                                                                     */
int gdmcf_debug_spmm_stamps(int n_waves, long long* host_out);
/* ---- graph step state: what changes from step to step, in device memory (hipGraph replay freezes kernel arguments) ----
 * A training step captured in a hipGraph (gdmcf_amd/graph.py) replays the SAME launches; the Philox offsets of the input
 * builder / timestep sampler and the AdamW bias corrections therefore come from a device block (opaque,
 * gdmcf_graph_state_bytes() bytes, filled on the host by gdmcf_graph_state_init and copied to the device by the caller)
 * that gdmcf_graph_state_tick advances by one step: offsets + 1, optimiser step + 1, scalars of that step taken from a
 * host-computed table (gdmcf_adam_hyper_fill: n entries of gdmcf_adam_hyper_bytes() bytes for steps first_step ..., the
 * same double-precision formulas as gdmcf_adamw_f32 -> bit-identical updates).  While a block is bound to the calling
 * thread (gdmcf_graph_state_bind; NULL unbinds) gdmcf_dnn_prep_input(_csr)_f32, gdmcf_sample_timesteps and gdmcf_adamw_*
 * ignore their offset / step arguments and read the block instead.                                                      */
int gdmcf_graph_state_bytes(void);
int gdmcf_adam_hyper_bytes(void);
int gdmcf_graph_state_init(void* state_host, uint64_t prep_offset, uint64_t ts_offset, int64_t adam_step, int64_t table_first,
                           int64_t table_len, const void* hyper_table_dev);
int gdmcf_adam_hyper_fill(void* out_host, int n, float lr, float beta1, float beta2, float eps, float weight_decay,
                          int64_t first_step, float grad_scale);
int gdmcf_graph_state_bind(const void* state_dev);
int gdmcf_graph_state_tick(void* state_dev, void* stream);
/* out = acc * scale */
int gdmcf_scale_f32(const float* acc, int64_t n, float scale, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDMCF_HIP_H */
