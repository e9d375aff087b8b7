// C-ABI entry points for the denoiser's dense layers: forward (plain / fused row-loss / fused
// posterior mean) and backward (input grad, weight grad).  All of them drive gemm_f32.hip.
#include <stdlib.h>

#include "common.h"

int gd_splitk_reduce(const float* slabs, int64_t slab_stride, int splits, int64_t ld_slab, int M, int N, int mode,
                     const float* bias, const float* rowscale, const float* aact, int64_t ldact, int act, float* out,
                     int64_t ldo, hipStream_t s);  // also refreshes the bf16 shadow of `out`, if one is registered
int gd_colsum(const float* dZ, int64_t ld, const float* rs, int M, int N, float* db, hipStream_t s);
int gd_rowpart_reduce(const float* rowpart, int ld, int M, int nt, float* rowsum, hipStream_t s);

namespace {

// workgroups aimed at by the split-K heuristic: 256 CUs x 2 resident, times ~1.5 so the second wave of
// workgroups evens out the tail (measured sweet spot on the batch-400 products; GDMCF_TARGET_WGS overrides)
static const int TARGET_WGS = getenv("GDMCF_TARGET_WGS") ? atoi(getenv("GDMCF_TARGET_WGS")) : 760;

// input precision of the dense products issued by the calling thread (gdmcf_gemm_precision)
thread_local int t_gemm_prec = GDMCF_GEMM_F32;

// tile class of an [M,N] output.  bf16 products are bound by L2 reads of the f32 sources, so batch-sized M takes
// the 208x256 class when that wastes little: <= 10 % row padding and, for products that cannot split K
// (`fused`: the epilogue needs complete sums), a last round of workgroups that is mostly full (one 120 KB-LDS
// workgroup per CU).
int pick_class(int M, int N, bool fused, int prec) {
    if (prec == GDMCF_GEMM_BF16 && M > 128) {
        const long pad = (long)gd_cdiv(M, 208) * 208;
        const long tiles = (pad / 208) * gd_cdiv(N, 256);
        const long rounds = (tiles + 255) / 256;
        if (pad * 10 <= (long)M * 11 && pad / 208 <= 2 && (!fused || tiles * 100 >= rounds * 256 * 85)) return 3;
    }
    // three-term split: 2.67x the matrix-pipe rate needs ~40 FLOP per operand byte from L2 -- batch-sized M takes 208-row
    // tiles (208x128: 39.6 FLOP/B against 24.6 at 80x128), one 129 KB-LDS workgroup per CU
    if (prec == GDMCF_GEMM_F32X3 && M > 128 && N >= 128) {
        const long pad = (long)gd_cdiv(M, 208) * 208;
        if (pad * 10 <= (long)M * 11) return 4;
    }
    return gd_pick_shape_class(M, N);
}

// weight-gradient products ([N_out x K_in] outputs, reduction over the batch): bf16 takes the 208x256 class when
// both dimensions are large and the last round of workgroups is mostly full (measured 0.397 -> 0.374 ms on the
// Amazon-Book shape; the kernel is bound by getting the f32 operands through L1, not by MFMA).
int pick_class_dw(int M, int N, int prec, bool fused_adamw = false) {
    static const int forced = getenv("GDMCF_BF16_DW_CLASS") ? atoi(getenv("GDMCF_BF16_DW_CLASS")) : -1;  // tuning knob
    if (prec == GDMCF_GEMM_BF16 && forced >= 0) return forced;
    // AdamW in the epilogue: the kernel is its optimiser stream (26 B per parameter) plus a short, latency-bound k loop that
    // nothing overlaps when one 208x256 workgroup owns the CU: three 80x128 workgroups per CU run one's k loop under the
    // others' streams (Amazon-Book shape, tools/bf16_fused_ablate.sh: 0.61 -> 0.55 ms per weight; 128x128: 0.59)
    if (prec == GDMCF_GEMM_BF16 && fused_adamw && M >= 160 && N >= 256) return 0;
    if (prec == GDMCF_GEMM_BF16 && M >= 416 && N >= 512) {
        const long tiles = (long)gd_cdiv(M, 208) * gd_cdiv(N, 256);
        const long rounds = (tiles + 255) / 256;
        if (tiles * 100 >= rounds * 256 * 85) return 3;
    }
    return gd_pick_shape_class(M, N);
}

// bf16 mode: stream the operands from their registered bf16 shadows when both have one of exactly this shape
// (operand stored [rows][K] for GD_LAY_KC, [K][rows] for GD_LAY_MC)
// result written by a fused epilogue (BIAS_ACT without split-K, LOSS): keep its bf16 shadow in sync
void attach_result_shadow(GdGemm& g) {
    GdShadow c;
    if (g.bf16 == 1 && gd_shadow_lookup(g.C, &c) && c.rows == g.M && c.cols >= g.N) {  // x_next lands in a wider xin buffer
        g.C16 = c.p16;
        g.ldc16 = c.ld16;
    }
}

void attach_shadows(GdGemm& g, int layA, int layB) {
    if (g.bf16 != 1) return;
    GdShadow a, b;
    if (!gd_shadow_lookup(g.A, &a) || !gd_shadow_lookup(g.B, &b)) return;
    const bool okA = layA == GD_LAY_KC ? (a.rows == g.M && a.cols == g.K) : (a.rows == g.K && a.cols == g.M);
    const bool okB = layB == GD_LAY_KC ? (b.rows == g.N && b.cols == g.K) : (b.rows == g.K && b.cols == g.N);
    if (!okA || !okB) return;
    g.A16 = a.p16; g.lda16 = a.ld16;
    g.B16 = b.p16; g.ldb16 = b.ld16;
}

// number of K splits for an [M,N,K] product whose output is small (batch x hidden)
int pick_splits(int M, int N, int K, int cls, int prec) {
    const int bk = prec == GDMCF_GEMM_BF16 ? 64 : 32;
    const int tiles = gd_cdiv(M, gd_gemm_tile_m(cls)) * gd_cdiv(N, gd_gemm_tile_n(cls));
    int s = (cls >= 3 ? 256 : TARGET_WGS) / tiles;  // classes 3 and 4: one workgroup per CU
    const int max_by_k = K / (8 * bk);  // keep >= 8 K-tiles per split
    if (s > max_by_k) s = max_by_k;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

inline int64_t round4(int64_t x) { return (x + 3) & ~(int64_t)3; }

// The bias gradient db[n] = sum_m rowscale[m] dZ[m, n] as column K of the weight-gradient product: possible when the caller
// states (a_scale_col) that column K of the activation operand A holds rowscale[m] -- gdmcf_rowscale_f32 writes it there when its
// output has room (ldo > K).  Nothing is remembered between calls.  Returns db when the request can be made, else NULL; a
// launcher that takes the request clears GdGemm::out2.
float* bias_col_request(int a_scale_col, int64_t lda, const float* rowscale, int K, float* db) {
    static const int on = getenv("GDMCF_BIAS_COL") ? atoi(getenv("GDMCF_BIAS_COL")) : 1;
    if (!on || !a_scale_col || db == nullptr || t_gemm_prec != GDMCF_GEMM_F32 || lda <= K) return nullptr;
    return db;
}

}  // namespace

extern "C" {

int gdmcf_gemm_precision(int mode) {
    const int prev = t_gemm_prec;
    if (mode == GDMCF_GEMM_F32 || mode == GDMCF_GEMM_BF16 || mode == GDMCF_GEMM_F32X3) t_gemm_prec = mode;
    return prev;
}

size_t gdmcf_linear_ws_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    // forward: slabs [splits][M][round4(N)];  backward-input: slabs [splits][M][round4(K)]
    size_t need = 0;
    for (int prec = GDMCF_GEMM_F32; prec <= GDMCF_GEMM_F32X3; ++prec) {  // the caller may switch precision later
        const int cf = pick_class(M, N, false, prec), cb = pick_class(M, K, false, prec);
        size_t f = (size_t)pick_splits(M, N, K, cf, prec) * M * round4(N);
        if (prec == GDMCF_GEMM_F32) {  // gdmcf_linear_fwd_wt_f32 on dr_kn_kernel
            const size_t fk = (size_t)gd_dr_kn_splits(M, N, K) * M * round4(N);
            f = f > fk ? f : fk;
        }
        size_t b = (size_t)pick_splits(M, K, N, cb, prec) * M * round4(K);
        if (prec == GDMCF_GEMM_F32) {  // the input gradient on dr_kn_kernel: one (split, tile) task per wave slot
            const size_t bk = (size_t)gd_dr_kn_splits(M, K, N) * M * round4(K);
            b = b > bk ? b : bk;
        }
        need = need > f ? need : f;
        need = need > b ? need : b;
    }
    return need * sizeof(float) + 256;
}

int gdmcf_loss_tiles(int N) { return gd_cdiv(N, 16); }  // partial row sums per output tile: the narrowest tile any kernel uses

int gdmcf_linear_fwd_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, int act, int M,
                         int N, int K, float* C, int64_t ldc, void* ws, size_t ws_bytes, void* stream) {
    GD_CHECK_SHAPE(M > 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldc >= N, "linear_fwd: bad shape");
    GD_CHECK_ARG(act == 0 || act == 1, "linear_fwd: bad activation");
    hipStream_t s = (hipStream_t)stream;
    const int cls = pick_class(M, N, false, t_gemm_prec);
    const int splits = pick_splits(M, N, K, cls, t_gemm_prec);
    GdGemm g = {};
    g.bf16 = t_gemm_prec;  // 0 f32, 1 bf16, 2 three-term split
    g.A = A; g.lda = lda; g.B = W; g.ldb = ldw; g.M = M; g.N = N; g.K = K;
    g.m_fastest = gd_cdiv(M, gd_gemm_tile_m(cls)) <= gd_cdiv(N, gd_gemm_tile_n(cls));
    g.bias = bias; g.act = act; g.prof_tag = 1;
    if (splits == 1) {
        g.splits = 1; g.C = C; g.ldc = ldc;
        attach_result_shadow(g);
        attach_shadows(g, GD_LAY_KC, GD_LAY_KC);
        return gd_gemm_launch(GD_LAY_KC, GD_LAY_KC, GD_EPI_BIAS_ACT, cls, g, s);
    }
    const int64_t lds_ = round4(N);
    const size_t need = (size_t)splits * M * lds_ * sizeof(float);
    if (ws == nullptr || ws_bytes < need) {
        gdmcf_set_error("linear_fwd: workspace %zu < %zu bytes", ws_bytes, need);
        return GDMCF_E_WORKSPACE;
    }
    g.splits = splits; g.C = (float*)ws; g.ldc = lds_; g.slab_stride = (int64_t)M * lds_;
    attach_shadows(g, GD_LAY_KC, GD_LAY_KC);
    int rc = gd_gemm_launch(GD_LAY_KC, GD_LAY_KC, GD_EPI_SLAB, cls, g, s);
    if (rc) return rc;
    const int real_splits = gd_cdiv(K, g.kchunk);
    return gd_splitk_reduce((const float*)ws, g.slab_stride, real_splits, lds_, M, N, 0, bias, nullptr, nullptr, 0,
                            act, C, ldc, s);
}

int gdmcf_linear_fwd_wt_f32(const float* A, int64_t lda, const float* Wt, int64_t ldwt, const float* bias, int act, int M, int N,
                            int K, float* C, int64_t ldc, void* ws, size_t ws_bytes, void* stream) {
    GD_CHECK_SHAPE(M > 0 && N > 0 && K > 0 && lda >= K && ldwt >= N && ldc >= N, "linear_fwd_wt: bad shape");
    GD_CHECK_ARG(act == 0 || act == 1, "linear_fwd_wt: bad activation");
    if (t_gemm_prec != GDMCF_GEMM_F32) {
        gdmcf_set_error("linear_fwd_wt: float32 GEMM mode only (the bf16 / f32x3 modes keep the weight in its own orientation)");
        return GDMCF_E_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    // the product of the input gradient, with a forward reducer: [M x N] = A[M x K] * Wt[K x N], split over K
    const int cls = pick_class(M, N, false, t_gemm_prec);
    const int splits = pick_splits(M, N, K, cls, t_gemm_prec);
    const int64_t lds_ = round4(N);
    const size_t need = (size_t)splits * M * lds_ * sizeof(float);
    if (ws == nullptr || ws_bytes < need) {
        gdmcf_set_error("linear_fwd_wt: workspace %zu < %zu bytes", ws_bytes, need);
        return GDMCF_E_WORKSPACE;
    }
    GdGemm g = {};
    g.A = A; g.lda = lda; g.B = Wt; g.ldb = ldwt; g.M = M; g.N = N; g.K = K;
    g.m_fastest = gd_cdiv(M, gd_gemm_tile_m(cls)) <= gd_cdiv(N, gd_gemm_tile_n(cls));
    g.splits = splits; g.C = (float*)ws; g.ldc = lds_; g.slab_stride = (int64_t)M * lds_; g.prof_tag = 1;
    g.ws_cap = ws_bytes;
    int rc = gd_gemm_launch(GD_LAY_KC, GD_LAY_MC, GD_EPI_SLAB, cls, g, s);
    if (rc) return rc;
    const int real_splits = gd_cdiv(K, g.kchunk);
    return gd_splitk_reduce((const float*)ws, g.slab_stride, real_splits, lds_, M, N, 0, bias, nullptr, nullptr, 0, act, C, ldc, s);
}

int gdmcf_linear_loss_fwd_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                              const float* target, int64_t ldt, const float* alpha, int M, int N, int K, float* out,
                              int64_t ldo, float* diff, int64_t ldd, float* rowpart, float* rowsum, void* stream) {
    GD_CHECK_SHAPE(M > 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldt >= N && ldd >= N, "linear_loss_fwd: bad shape");
    GD_CHECK_SHAPE(out == nullptr || ldo >= N, "linear_loss_fwd: ldo < N");
    hipStream_t s = (hipStream_t)stream;
    const int cls = pick_class(M, N, true, t_gemm_prec);
    GdGemm g = {};
    g.bf16 = t_gemm_prec;  // 0 f32, 1 bf16, 2 three-term split
    g.A = A; g.lda = lda; g.B = W; g.ldb = ldw; g.M = M; g.N = N; g.K = K; g.splits = 1;
    g.m_fastest = gd_cdiv(M, gd_gemm_tile_m(cls)) <= gd_cdiv(N, gd_gemm_tile_n(cls));
    g.bias = bias; g.aux = target; g.ldaux = ldt; g.r0 = alpha; g.out2 = out; g.ldout2 = ldo;
    g.prof_tag = 2;
    g.C = diff; g.ldc = ldd; g.rowpart = rowpart; g.ld_rowpart = gdmcf_loss_tiles(N);
    attach_result_shadow(g);
    attach_shadows(g, GD_LAY_KC, GD_LAY_KC);
    int rc = gd_gemm_launch(GD_LAY_KC, GD_LAY_KC, GD_EPI_LOSS, cls, g, s);
    if (rc) return rc;
    return gd_rowpart_reduce(rowpart, g.ld_rowpart, M, g.tiles_n, rowsum, s);
}

int gdmcf_linear_loss_fwd_bits_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                                   const uint32_t* target_bits, int64_t ldbits, const float* alpha, int M, int N, int K,
                                   float* out, int64_t ldo, float* diff, int64_t ldd, float* rowpart, float* rowsum,
                                   void* stream) {
    GD_CHECK_SHAPE(M > 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldd >= N, "linear_loss_fwd_bits: bad shape");
    GD_CHECK_SHAPE(out == nullptr || ldo >= N, "linear_loss_fwd_bits: ldo < N");
    GD_CHECK_ARG(target_bits && ldbits >= (N + 31) / 32, "linear_loss_fwd_bits: target bitmap missing / ldbits < ceil(N/32)");
    hipStream_t s = (hipStream_t)stream;
    const int cls = pick_class(M, N, true, t_gemm_prec);
    GdGemm g = {};
    g.bf16 = t_gemm_prec;  // 0 f32, 1 bf16, 2 three-term split
    g.A = A; g.lda = lda; g.B = W; g.ldb = ldw; g.M = M; g.N = N; g.K = K; g.splits = 1;
    g.m_fastest = gd_cdiv(M, gd_gemm_tile_m(cls)) <= gd_cdiv(N, gd_gemm_tile_n(cls));
    g.bias = bias; g.aux = nullptr; g.ldaux = 0; g.aux_bits = target_bits; g.ldbits = ldbits; g.r0 = alpha; g.out2 = out;
    g.ldout2 = ldo; g.prof_tag = 2;
    g.C = diff; g.ldc = ldd; g.rowpart = rowpart; g.ld_rowpart = gdmcf_loss_tiles(N);
    attach_result_shadow(g);
    attach_shadows(g, GD_LAY_KC, GD_LAY_KC);
    int rc = gd_gemm_launch(GD_LAY_KC, GD_LAY_KC, GD_EPI_LOSS, cls, g, s);
    if (rc) return rc;
    return gd_rowpart_reduce(rowpart, g.ld_rowpart, M, g.tiles_n, rowsum, s);
}

int gdmcf_linear_posterior_fwd_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                                   const float* x_t, int64_t ldxt, const float* c1, const float* c2, const float* r1,
                                   const float* r2, const float* sigma, const float* z, int64_t ldz, int M, int N,
                                   int K, float* x_next, int64_t ldxn, float* pred_out, int64_t ldp, void* stream) {
    GD_CHECK_SHAPE(M > 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldxt >= N && ldxn >= N, "linear_posterior_fwd: bad shape");
    GD_CHECK_ARG(c1 && c2, "linear_posterior_fwd: c1/c2 missing");
    GD_CHECK_ARG((r1 == nullptr) == (r2 == nullptr), "linear_posterior_fwd: r1/r2 must both be set or both NULL");
    GD_CHECK_ARG(z == nullptr || (sigma != nullptr && ldz >= N), "linear_posterior_fwd: sigma missing");
    hipStream_t s = (hipStream_t)stream;
    const int cls = pick_class(M, N, true, t_gemm_prec);
    GdGemm g = {};
    g.bf16 = t_gemm_prec;  // 0 f32, 1 bf16, 2 three-term split
    g.A = A; g.lda = lda; g.B = W; g.ldb = ldw; g.M = M; g.N = N; g.K = K; g.splits = 1;
    g.m_fastest = gd_cdiv(M, gd_gemm_tile_m(cls)) <= gd_cdiv(N, gd_gemm_tile_n(cls));
    g.bias = bias; g.aux = x_t; g.ldaux = ldxt; g.aux2 = z; g.ldaux2 = ldz;
    g.r0 = c1; g.r1 = c2; g.r2 = r1; g.r3 = r2; g.r4 = sigma;
    g.out2 = pred_out; g.ldout2 = ldp; g.C = x_next; g.ldc = ldxn; g.prof_tag = 3;
    attach_result_shadow(g);
    attach_shadows(g, GD_LAY_KC, GD_LAY_KC);
    return gd_gemm_launch(GD_LAY_KC, GD_LAY_KC, GD_EPI_POST, cls, g, s);
}

int gdmcf_linear_bwd_input_f32(const float* dZ, int64_t lddz, const float* W, int64_t ldw, const float* rowscale,
                               const float* Aact, int64_t ldact, int act, int M, int N, int K, float* dA,
                               int64_t ldda, void* ws, size_t ws_bytes, void* stream) {
    GD_CHECK_SHAPE(M > 0 && N > 0 && K > 0 && lddz >= N && ldw >= K && ldda >= K, "linear_bwd_input: bad shape");
    GD_CHECK_ARG(act == 0 || (act == 1 && Aact && ldact >= K), "linear_bwd_input: activation output missing");
    hipStream_t s = (hipStream_t)stream;
    // product dims: [M x K_in] = dZ[M x N] * W[N x K_in]  -> gemm (M, K, reduction N)
    const int cls = pick_class(M, K, false, t_gemm_prec);
    const int splits = pick_splits(M, K, N, cls, t_gemm_prec);
    const int64_t lds_ = round4(K);
    const size_t need = (size_t)splits * M * lds_ * sizeof(float);
    if (ws == nullptr || ws_bytes < need) {
        gdmcf_set_error("linear_bwd_input: workspace %zu < %zu bytes", ws_bytes, need);
        return GDMCF_E_WORKSPACE;
    }
    GdGemm g = {};
    g.bf16 = t_gemm_prec;  // 0 f32, 1 bf16, 2 three-term split
    g.A = dZ; g.lda = lddz; g.B = W; g.ldb = ldw; g.M = M; g.N = K; g.K = N;
    g.m_fastest = gd_cdiv(M, gd_gemm_tile_m(cls)) <= gd_cdiv(K, gd_gemm_tile_n(cls));
    g.splits = splits; g.C = (float*)ws; g.ldc = lds_; g.slab_stride = (int64_t)M * lds_; g.prof_tag = 4;
    g.ws_cap = ws_bytes;  // (the register-streaming kernel splits further when the workspace allows: gdmcf_linear_ws_bytes sizes for it)
    attach_shadows(g, GD_LAY_KC, GD_LAY_MC);
    int rc = gd_gemm_launch(GD_LAY_KC, GD_LAY_MC, GD_EPI_SLAB, cls, g, s);
    if (rc) return rc;
    const int real_splits = gd_cdiv(N, g.kchunk);
    return gd_splitk_reduce((const float*)ws, g.slab_stride, real_splits, lds_, M, K, 1, nullptr, rowscale, Aact,
                            ldact, act, dA, ldda, s);
}

int gdmcf_linear_bwd_weight_f32(const float* dZ, int64_t lddz, const float* A, int64_t lda, const float* rowscale,
                                int a_scale_col, int M, int N, int K, float* dW, int64_t lddw, float* db, int accumulate,
                                void* stream) {
    GD_CHECK_SHAPE(M > 0 && N > 0 && K > 0 && lddz >= N && lda >= K && lddw >= K, "linear_bwd_weight: bad shape");
    hipStream_t s = (hipStream_t)stream;
    // dW[N x K_in] = dZ[M x N]^T * A[M x K_in]  -> gemm (N, K, reduction M), both operands row-contiguous
    const int cls = pick_class_dw(N, K, t_gemm_prec);
    GdGemm g = {};
    g.bf16 = t_gemm_prec;  // 0 f32, 1 bf16, 2 three-term split
    g.A = dZ; g.lda = lddz; g.B = A; g.ldb = lda; g.M = N; g.N = K; g.K = M; g.splits = 1;
    g.m_fastest = gd_cdiv(N, gd_gemm_tile_m(cls)) <= gd_cdiv(K, gd_gemm_tile_n(cls));
    g.C = dW; g.ldc = lddw; g.accumulate = accumulate; g.prof_tag = 5;
    attach_shadows(g, GD_LAY_MC, GD_LAY_MC);
    g.out2 = bias_col_request(a_scale_col, lda, rowscale, K, db);
    const bool asked = g.out2 != nullptr;
    int rc = gd_gemm_launch(GD_LAY_MC, GD_LAY_MC, GD_EPI_STORE, cls, g, s);
    if (rc) return rc;
    if (asked && g.out2 == nullptr) return GDMCF_OK;  // db came out of the product
    if (db) return gd_colsum(dZ, lddz, rowscale, M, N, db, s);
    return GDMCF_OK;
}

int gdmcf_linear_bwd_weight_adamw_f32(const float* dZ, int64_t lddz, const float* A, int64_t lda, const float* rowscale,
                                      int a_scale_col, int M, int N, int K, float* W, int64_t ldw, float* exp_avg, float* exp_avg_sq,
                                      float* db, float lr, float beta1, float beta2, float eps, float weight_decay,
                                      int step, float grad_scale, void* stream) {
    GD_CHECK_SHAPE(M > 0 && N > 0 && K > 0 && lddz >= N && lda >= K && ldw >= K, "linear_bwd_weight_adamw: bad shape");
    GD_CHECK_ARG(W && exp_avg && exp_avg_sq && step >= 1, "linear_bwd_weight_adamw: optimizer state missing");
    hipStream_t s = (hipStream_t)stream;
    const int cls = pick_class_dw(N, K, t_gemm_prec, true);
    GdGemm g = {};
    g.bf16 = t_gemm_prec;  // 0 f32, 1 bf16, 2 three-term split
    g.A = dZ; g.lda = lddz; g.B = A; g.ldb = lda; g.M = N; g.N = K; g.K = M; g.splits = 1;
    g.m_fastest = gd_cdiv(N, gd_gemm_tile_m(cls)) <= gd_cdiv(K, gd_gemm_tile_n(cls));
    g.C = W; g.ldc = ldw; g.aux = exp_avg; g.aux2 = exp_avg_sq; g.prof_tag = 5;
    g.adam = gd_adam_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
    g.adam_dev = t_gd_step_state ? &t_gd_step_state->hyper : nullptr;  // bound graph step state: the scalars of the step being replayed
    attach_result_shadow(g);  // the bf16 kernels' row epilogue also refreshes W's bf16 shadow
    attach_shadows(g, GD_LAY_MC, GD_LAY_MC);
    g.out2 = bias_col_request(a_scale_col, lda, rowscale, K, db);
    const bool asked = g.out2 != nullptr;
    int rc = gd_gemm_launch(GD_LAY_MC, GD_LAY_MC, GD_EPI_ADAMW, cls, g, s);
    if (rc) return rc;
    if (asked && g.out2 == nullptr) return GDMCF_OK;  // db came out of the product
    if (db) return gd_colsum(dZ, lddz, rowscale, M, N, db, s);
    return GDMCF_OK;
}

}  // extern "C"
