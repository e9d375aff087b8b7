// Shared declarations for libgdmcf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gdmcf_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

void gdmcf_set_error(const char* fmt, ...);

#define GD_CHECK_SHAPE(cond, msg)            \
    do {                                     \
        if (!(cond)) {                       \
            gdmcf_set_error("%s", msg);      \
            return GDMCF_E_SHAPE;            \
        }                                    \
    } while (0)

#define GD_CHECK_ARG(cond, msg)              \
    do {                                     \
        if (!(cond)) {                       \
            gdmcf_set_error("%s", msg);      \
            return GDMCF_E_ARG;              \
        }                                    \
    } while (0)

static inline int gd_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gdmcf_set_error("%s: %s", what, hipGetErrorString(e));
        return GDMCF_E_HIP;
    }
    return GDMCF_OK;
}

// ---- optional event timing (capi.hip) ----------------------------------------------------
extern bool g_gd_prof_on;
void gd_prof_begin(int tag, double work, hipStream_t s);
void gd_prof_end(hipStream_t s);
struct GdProfScope {
    hipStream_t s;
    bool on;
    GdProfScope(int tag, double work, hipStream_t st) : s(st), on(g_gd_prof_on) {
        if (on) gd_prof_begin(tag, work, s);
    }
    ~GdProfScope() {
        if (on) gd_prof_end(s);
    }
};

static inline int gd_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline bool gd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- bf16 shadow registry (capi.hip) ----------------------------------------------------------------------
struct GdShadow {
    void* p16;
    int64_t ld16;
    int64_t rows, cols;
};
bool gd_shadow_lookup(const void* f32, GdShadow* out);
int gd_cast_bf16(const float* src, int64_t ld, void* dst, int64_t ld16, int64_t rows, int64_t cols, hipStream_t s);

// ---- GEMM core (gemm_f32.hip) -----------------------------------------------------------
enum { GD_LAY_KC = 0, GD_LAY_MC = 1 };  // operand stored [rows][K] (K contiguous) / [K][rows]
enum { GD_EPI_SLAB = 0, GD_EPI_BIAS_ACT = 1, GD_EPI_LOSS = 2, GD_EPI_POST = 3, GD_EPI_STORE = 4, GD_EPI_ADAMW = 5 };

// AdamW scalars (torch.optim.AdamW single-tensor math), shared by the stand-alone kernel and the fused epilogue
struct GdAdamHyper {
    float decay;     // 1 - lr*wd
    float one_m_b1;  // 1 - beta1
    float beta2;
    float one_m_b2;
    float bc2_sqrt;  // sqrt(1 - beta2^step)
    float eps;
    float neg_step;  // -lr / (1 - beta1^step)
    float grad_scale;
    float inv_bc2_sqrt;  // 1 / sqrt(1 - beta2^step), from the double
    float pad_[3];       // (16-byte multiple: the struct is copied to the device in tables)
};
GdAdamHyper gd_adam_hyper(float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale);

// One element of torch.optim.AdamW's single-tensor update (torch/optim/adamw.py; reference main.py:258, :351).
// Every kernel that updates parameters inlines THIS function (the stand-alone pass, the epilogues of the LDS-tiled products, the
// optimiser stream of the register-streaming product -- whose row groups go through one of three code paths depending on which
// wave picks a tile when): its rounding must not depend on the instance, or results would differ from run to run.  Hence no
// compiler contraction (the fused multiply-adds are written out), and the two divisions and the square root of the step --
// p -= step_size * m / (sqrt(v) / sqrt(bc2) + eps) -- as hardware reciprocal / square root (1 ulp each) instead of the correctly
// rounded sequences (~45 of the update's ~60 instructions).  v_mfma_f32_16x16x4_f32 runs at the vector rate ON the vector
// ALUs' issue port (measured round 4: arithmetic placed beside the MFMAs of a SIMD's other wave does not overlap, it adds), so in
// the fused product those instructions are matrix time lost.  The step term is then within ~3e-7 (relative) of the correctly
// rounded one, i.e. within 1e-12 absolute at lr = 1e-5: below half an ulp of any weight it is added to, except on rounding
// ties.  exp_avg and exp_avg_sq are exactly the reference's fused-multiply-add forms.
__device__ __forceinline__ void gd_adam_elem(float& p, float g, float& m, float& v, const GdAdamHyper& h) {
#pragma clang fp contract(off)
    g = g * h.grad_scale;
    p = p * h.decay;
    m = __builtin_fmaf(g - m, h.one_m_b1, m);
    v = __builtin_fmaf(h.one_m_b2 * g, g, v * h.beta2);
    const float denom = __builtin_fmaf(__builtin_amdgcn_sqrtf(v), h.inv_bc2_sqrt, h.eps);
    p = __builtin_fmaf(h.neg_step * m, __builtin_amdgcn_rcpf(denom), p);
}

// ---- graph step state (kernels_misc.hip: gdmcf_graph_state_*) ---------------------------------------------------------
// When a training step is captured in a hipGraph its kernel ARGUMENTS are frozen; what changes from step to step (the
// Philox offsets of the input builder and of the timestep sampler, the AdamW bias corrections) therefore lives in device
// memory: a block of this layout, advanced by gdmcf_graph_state_tick at the start of every step (eager or replayed).
struct GdStepState {
    uint64_t prep_offset;  // Philox offset of the next input-builder launch
    uint64_t ts_offset;    // ... of the next timestep draw
    int64_t adam_step;     // optimiser steps taken
    int64_t table_first;   // hyper_table[k] belongs to optimiser step table_first + k
    int64_t table_len;
    const GdAdamHyper* hyper_table;  // host-computed scalars of the coming steps (bit-identical to the eager path)
    GdAdamHyper hyper;               // the current step's
};
extern thread_local const GdStepState* t_gd_step_state;  // bound by gdmcf_graph_state_bind; NULL = by-value arguments

struct GdGemm {
    const float* A;
    int64_t lda;
    const float* B;
    int64_t ldb;
    int M, N, K;
    int kchunk;  // K range per split (multiple of the kernel's BK)
    int splits;
    int tiles_m, tiles_n;
    int m_fastest;  // tile order inside one split
    float* C;
    int64_t ldc;
    int64_t slab_stride;
    const float* bias;
    int act;
    const float* aux;  // target (LOSS) / x_t (POST)
    int64_t ldaux;
    const uint32_t* aux_bits;  // LOSS: the target as a bitmap of {0,1} rows instead of aux (word n>>5 of row m, bit n&31)
    int64_t ldbits;
    const float* aux2;  // z noise (POST)
    int64_t ldaux2;
    const float* r0;  // LOSS: alpha[m];  POST: c1[m]
    const float* r1;  // POST: c2
    const float* r2;  // POST: r1 (eps) or NULL
    const float* r3;  // POST: r2 (eps)
    const float* r4;  // POST: sigma
    float* out2;  // LOSS: raw model output;  POST: pred_xstart;  STORE / ADAMW on the register-streaming kernel: bias gradient
                  // [M] taken from column N of the product (operand B carries one more column); cleared by the kernel's launcher
                  // when it has taken the request
    int64_t ldout2;
    float* rowpart;
    int ld_rowpart;
    int accumulate;
    int prof_tag;
    GdAdamHyper adam;  // GD_EPI_ADAMW: C = parameter, aux = exp_avg, aux2 = exp_avg_sq (all [M,N], ldc)
    const GdAdamHyper* adam_dev;  // ... this step's scalars in device memory instead (a bound graph step state: the step replayed from a
                                  // hipGraph reads them from the block its tick kernel advances); NULL = `adam`
    // bf16 mode only: bf16 copies ("shadows", gdmcf_bf16_shadow_set) of the operands / of the LOSS epilogue's
    // result.  When BOTH operand shadows are present the kernel streams them instead of the f32 matrices.
    const void* A16;
    int64_t lda16;
    const void* B16;
    int64_t ldb16;
    void* C16;
    int64_t ldc16;
    size_t ws_cap;  // GD_EPI_SLAB: bytes of workspace behind C (0: unknown) -- a launcher that wants more splits than the caller sized checks it
    int dbg;   // timing ablations of gemm_split.hip (GDMCF_SPLIT_DBG; 0 in production)
    int bf16;  // 1: operands rounded to bfloat16 on the way to LDS, bf16 MFMA, f32 accumulate (gemm_bf16.hip)
               // 2: operands split into three bfloat16 terms, six bf16 MFMAs per product block (gemm_split.hip)
};

// Split count the register-streaming input-gradient kernel (csrc/gemm_dr.hip: dr_kn_kernel) would use for C[M, N] = A[M, K] B[K, N],
// or 0 when it does not take the product: gdmcf_linear_ws_bytes sizes the slab workspace with it.
int gd_dr_kn_splits(int M, int N, int K);
extern thread_local int t_gd_last_gemm;  // gdmcf_debug_last_gemm (include/gdmcf_hip.h)

// shape classes: 0 = "batch-M" (BM=80, BN=128), 1 = square 128x128, 2 = small 64x64,
//                3 = bf16 only: 208x256 on 8 waves (batch-sized M, see gemm_bf16.hip)
int gd_gemm_launch(int layA, int layB, int epi, int shape_class, GdGemm& g, hipStream_t s);
int gd_gemm_bf16_launch(int layA, int layB, int epi, int shape_class, GdGemm& g, hipStream_t s);  // g.bf16 != 0
int gd_gemm_split_launch(int layA, int layB, int epi, int shape_class, GdGemm& g, hipStream_t s);  // g.bf16 == 2 (gemm_split.hip)
int gd_gemm_small_launch(int layA, int layB, int epi, GdGemm& g, hipStream_t s);  // degenerate shapes (gemm_small.hip)
// direct-to-register f32 products (gemm_dr.hip): GD_DR_NOT_TAKEN = not a product / shape it handles, fall back to the LDS-tiled kernels
enum { GD_DR_NOT_TAKEN = 1 };
int gd_gemm_dr_launch(int layA, int layB, int epi, GdGemm& g, hipStream_t s);
int gd_gemm_tile_m(int shape_class);
int gd_gemm_tile_n(int shape_class);
int gd_gemm_bk(int layA, int layB);
int gd_pick_shape_class(int M, int N);
