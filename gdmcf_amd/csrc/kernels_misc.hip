// HBM-bound kernels of the diffusion hot path: denoiser-input builder (q_sample + normalize +
// dropout + timestep embedding + cat), split-K reducers, bias/embedding gradients, the float64
// per-row loss tail with the Lt-history FIFO, and the fused multi-tensor AdamW.
#include <math.h>

#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (counter-based: any element can be regenerated in any kernel, so noise and
// dropout masks never have to be stored).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        // (one 32 x 32 -> 64 product per word pair: v_mad_u64_u32 where hipcc picks it -- half the quarter-rate multiplies of a
        // v_mul_lo_u32 / v_mul_hi_u32 pair)
        const uint64_t p0 = (uint64_t)M0 * (uint64_t)c.x, p1 = (uint64_t)M1 * (uint64_t)c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0;
        k.y += W1;
    }
    return c;
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = ((float)a + 1.0f) * 2.3283064365386963e-10f;  // (0,1]
    const float u2 = (float)b * 2.3283064365386963e-10f;
    // hardware transcendentals (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32 take the angle in revolutions): the libm
    // forms cost ~10x the instructions and made the input builder ALU-bound; ~1e-6 absolute error is irrelevant for
    // N(0,1) noise (tests/test_gpu_parity.py::test_philox_noise_and_dropout_statistics)
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u1)
    z0 = rad * __builtin_amdgcn_cosf(u2);
    z1 = rad * __builtin_amdgcn_sinf(u2);
}

// round-to-nearest-even float -> bfloat16 bits (v_cvt_pk_bf16_f32), for the bf16 shadow copies
__device__ __forceinline__ unsigned short gd_bf16_bits(float x) {
    __bf16 h = (__bf16)x;
    return __builtin_bit_cast(unsigned short, h);
}

struct PrepArgs {
    const float* x;
    int64_t ldx;
    const int64_t* ts;
    const float* ca;
    const float* cb;
    int noise_mode;
    const float* noise;
    int64_t ldn;
    int drop_mode;
    const uint8_t* keep;
    int64_t ldkeep;
    float drop_scale;  // 1/(1-p)
    uint32_t keep_thresh;  // Philox: keep iff (16-bit uniform) < keep_thresh = round((1 - p) * 65536): the keep probability is
                           // quantised to 2^-16 (exact for p = 0.5); ONE block gives the 8 uniforms of two column groups
    uint64_t seed, offset;
    const GdStepState* step_state;  // graph mode: the Philox offset is read from the device (NULL: `offset`)
    const float* rownorm;  // [B] L2 norms of x_t rows (normalize) or NULL
    const float* emb_w;
    const float* emb_b;
    int E, B, I;
    float* xin;
    int64_t ldxin;
    float* xt_out;
    int64_t ldxt;
    float* temb_out;
    unsigned short* xin16;  // bf16 shadow of xin (or NULL), row stride ldxin16 (a multiple of 64 >= I+E)
    int64_t ldxin16;
    // CSR source (gdmcf_dnn_prep_input_csr_f32): row b of the batch is row csr_rows[b] of a {0,1} matrix held as CSR;
    // x is NULL then.  bits_out receives the rows as bitmaps (word w of row b = columns 32w .. 32w+31), the loss target.
    const int64_t* csr_indptr;
    const int32_t* csr_indices;
    const int64_t* csr_rows;
    uint32_t* bits_out;
    int64_t ldbits;
};

// (out of line: a few hundred threads of a launch evaluate it, but inlined its libm sinusoids -- Payne-Hanek reductions and all --
// were two thirds of the input builder's 8 500 lines of ISA and cost the hot loop ~3.5 us of instruction fetch,
// profiles/r04_prep_input_ablation.txt)
__device__ __noinline__ float temb_value(float t, int f, int E) {
    // reference models/DNN.py:1817-1825: [cos(t*freqs), sin(t*freqs), (0 if E odd)]
    const int half = E / 2;
    if (f >= 2 * half) return 0.f;
    const int j = (f < half) ? f : f - half;
    const float freq = expf(-9.210340371976184f * (float)j / (float)half);
    const float a = t * freq;
    return (f < half) ? cosf(a) : sinf(a);
}

// x_t for 4 consecutive columns of one row (shared by the row-norm pass and the main pass)
template <bool FULL = false>  // FULL: the caller knows col + 3 < I (no per-element bounds checks)
__device__ __forceinline__ void xt4(const PrepArgs& a, int b, int col, float ca, float cb, float (&v)[4],
                                    const uint32_t* bm = nullptr, int bm_col0 = 0) {
    const float* xr = a.x + (int64_t)b * a.ldx;
    if (bm) {  // CSR source: the workgroup's span as a bitmap in LDS (col is a multiple of 4: one word holds all four)
        const uint32_t w = bm[(col - bm_col0) >> 5] >> ((col - bm_col0) & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((FULL || col + j < a.I) && ((w >> j) & 1u)) ? 1.f : 0.f;
    } else if (FULL || col + 3 < a.I) {
        // one 16-byte load (rows of the dense batch are only 4-byte aligned when I is odd: gfx950 takes that)
        typedef f32x4 f32x4_u4 __attribute__((aligned(4)));
        const f32x4 t4 = *reinterpret_cast<const f32x4_u4*>(xr + col);
        v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (col + j < a.I) ? xr[col + j] : 0.f;
    }
    if (a.ca) {
        float nz[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.noise_mode == 1) {
            const float* nr = a.noise + (int64_t)b * a.ldn;
            if (FULL || col + 3 < a.I) {
                typedef f32x4 f32x4_u4 __attribute__((aligned(4)));
                const f32x4 t4 = *reinterpret_cast<const f32x4_u4*>(nr + col);
                nz[0] = t4.x; nz[1] = t4.y; nz[2] = t4.z; nz[3] = t4.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (col + j < a.I) nz[j] = nr[col + j];
            }
        } else if (a.noise_mode == 2) {
            const uint4 r = philox4x32_10(make_uint4((uint32_t)(col >> 2), (uint32_t)b, 0u, (uint32_t)a.offset),
                                          make_uint2((uint32_t)a.seed, (uint32_t)(a.seed >> 32)));
            box_muller(r.x, r.y, nz[0], nz[1]);
            box_muller(r.z, r.w, nz[2], nz[3]);
        }
#pragma unroll
        // two rounded products + one rounded sum, exactly as the reference's mul, mul, add (:403-407):
        // no FMA contraction here (HIP's __fmul_rn/__fadd_rn are plain operators and would still fuse)
        for (int j = 0; j < 4; ++j) {
#pragma clang fp contract(off)
            const float p0 = ca * v[j];
            const float p1 = cb * nz[j];
            v[j] = p0 + p1;
        }
    }
}

__global__ __launch_bounds__(256) void prep_rowss_kernel(PrepArgs a, float* __restrict__ rownorm) {
    if (a.step_state) a.offset = a.step_state->prep_offset;
    const int b = blockIdx.x;
    float ca = 1.f, cb = 0.f;
    if (a.ca) {
        const int64_t t = a.ts[b];
        ca = a.ca[t];
        cb = a.cb[t];
    }
    float ss = 0.f;
    for (int col = threadIdx.x * 4; col < a.I; col += 256 * 4) {
        float v[4];
        xt4(a, b, col, ca, cb, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) ss += v[j] * v[j];
    }
    __shared__ float red[4];
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) rownorm[b] = sqrtf(red[0] + red[1] + red[2] + red[3]);
}

// PREP_G column groups of 4 per thread (strided by the workgroup's 1024 columns): the per-row scalars (ts, coefficients)
// are fetched once per thread and each workgroup moves 16 KB in and out instead of 4 KB.
constexpr int PREP_G = 4;

__global__ __launch_bounds__(256) void prep_input_kernel(PrepArgs a) {
    if (a.step_state) a.offset = a.step_state->prep_offset;
    const int b = blockIdx.y;
    const int64_t t = a.ts ? a.ts[b] : 0;
    float ca = 1.f, cb = 0.f;
    if (a.ca) {
        ca = a.ca[t];
        cb = a.cb[t];
    }
    // the workgroup that holds the embedding columns evaluates the E sinusoids ONCE, one per lane, instead of E times
    // per embedding column in a serial chain of libm calls (that chain was a ~15 us tail of the whole launch)
    __shared__ float s_temb[256];
    __shared__ uint32_t s_bm[256 * PREP_G * 4 / 32];  // CSR source: this workgroup's 4096 columns of row b as bits
    const int bm_col0 = blockIdx.x * (256 * PREP_G * 4);
    if (a.csr_indptr) {
        if (threadIdx.x < 256 * PREP_G * 4 / 32) s_bm[threadIdx.x] = 0u;
        __syncthreads();
        const int64_t r = a.csr_rows[b];
        const int64_t beg = a.csr_indptr[r], end = a.csr_indptr[r + 1];
        for (int64_t k = beg + threadIdx.x; k < end; k += 256) {
            const int c = a.csr_indices[k] - bm_col0;
            if (c >= 0 && c < 256 * PREP_G * 4) atomicOr(&s_bm[c >> 5], 1u << (c & 31));
        }
        __syncthreads();
        if (a.bits_out && threadIdx.x < 256 * PREP_G * 4 / 32) {
            const int64_t w = (int64_t)(bm_col0 >> 5) + threadIdx.x;
            if (w < a.ldbits) a.bits_out[(int64_t)b * a.ldbits + w] = s_bm[threadIdx.x];
        }
    }
    const uint32_t* bm = a.csr_indptr ? s_bm : nullptr;
    const bool has_emb = a.E > 0 && a.E <= 256 && (int)((blockIdx.x + 1) * (256 * PREP_G * 4)) > a.I;
    if (has_emb) {
        if ((int)threadIdx.x < a.E) s_temb[threadIdx.x] = temb_value((float)t, threadIdx.x, a.E);
        __syncthreads();
    }
    const uint2 key = make_uint2((uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const int col_base = (blockIdx.x * (256 * PREP_G) + threadIdx.x) * 4;  // group u of this thread: col_base + 1024 u
    // ---- the hot path: whole groups of four item columns (all but the last group or two of a row) ----
    uint4 dr = make_uint4(0u, 0u, 0u, 0u);  // dropout uniforms of a PAIR of column groups (u, u + 1): 16 bits per element
#pragma unroll
    for (int u = 0; u < PREP_G; ++u) {
        const int col = col_base + u * 1024;
        if (a.drop_mode == 2 && (u & 1) == 0 && col < a.I)
            dr = philox4x32_10(make_uint4((uint32_t)(col >> 2), (uint32_t)b, 1u, (uint32_t)a.offset), key);
        if (col + 3 >= a.I) continue;  // (the tail pass below)
        const int dsh = 16 * (u & 1);
        const uint32_t du[4] = {(dr.x >> dsh) & 0xFFFFu, (dr.y >> dsh) & 0xFFFFu, (dr.z >> dsh) & 0xFFFFu, (dr.w >> dsh) & 0xFFFFu};
        float v[4];
        xt4<true>(a, b, col, ca, cb, v, bm, bm_col0);
        if (a.xt_out) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a.xt_out[(int64_t)b * a.ldxt + col + j] = v[j];
        }
        if (a.rownorm) {
            const float dn = fmaxf(a.rownorm[b], 1e-12f);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] / dn;
        }
        if (a.drop_mode == 1) {
            const uint8_t* kr = a.keep + (int64_t)b * a.ldkeep + col;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = kr[j] ? v[j] * a.drop_scale : 0.f;
        } else if (a.drop_mode == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (du[j] < a.keep_thresh) ? v[j] * a.drop_scale : 0.f;
        }
        *reinterpret_cast<f32x4*>(a.xin + (int64_t)b * a.ldxin + col) = f32x4{v[0], v[1], v[2], v[3]};
        if (a.xin16 && col < a.ldxin16) {
            const uint2 w = make_uint2(gd_bf16_bits(v[0]) | ((unsigned)gd_bf16_bits(v[1]) << 16),
                                       gd_bf16_bits(v[2]) | ((unsigned)gd_bf16_bits(v[3]) << 16));
            *reinterpret_cast<uint2*>(a.xin16 + (int64_t)b * a.ldxin16 + col) = w;
        }
    }
    // ---- the tail pass: the row's last (ragged) item group, the timestep-embedding columns [I, I+E) and the zero padding up to
    // ldxin -- a few dozen groups of a row, ONE rolled instance of the code (unrolled beside the hot path it was most of the
    // kernel's 8 500 lines of ISA: ~3.5 us of instruction fetch, profiles/r04_prep_input_ablation.txt) ----
#pragma unroll 1
    for (int u = 0; u < PREP_G; ++u) {
        const int col = col_base + u * 1024;
        if (col + 3 < a.I || col >= a.ldxin) continue;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (col < a.I) {
            xt4(a, b, col, ca, cb, v, bm, bm_col0);
            if (a.xt_out) {
                for (int j = 0; j < 4; ++j)
                    if (col + j < a.I) a.xt_out[(int64_t)b * a.ldxt + col + j] = v[j];
            }
            if (a.rownorm) {
                const float dn = fmaxf(a.rownorm[b], 1e-12f);
                for (int j = 0; j < 4; ++j) v[j] = v[j] / dn;
            }
            if (a.drop_mode == 1) {
                const uint8_t* kr = a.keep + (int64_t)b * a.ldkeep;
                for (int j = 0; j < 4; ++j)
                    if (col + j < a.I) v[j] = kr[col + j] ? v[j] * a.drop_scale : 0.f;
            } else if (a.drop_mode == 2) {  // the pair's block again (same counter as in the hot path: group u & ~1 of this thread)
                const uint4 d2 = philox4x32_10(make_uint4((uint32_t)((col - (u & 1) * 1024) >> 2), (uint32_t)b, 1u, (uint32_t)a.offset), key);
                const int dsh = 16 * (u & 1);
                const uint32_t du[4] = {(d2.x >> dsh) & 0xFFFFu, (d2.y >> dsh) & 0xFFFFu, (d2.z >> dsh) & 0xFFFFu, (d2.w >> dsh) & 0xFFFFu};
                for (int j = 0; j < 4; ++j) v[j] = (du[j] < a.keep_thresh) ? v[j] * a.drop_scale : 0.f;
            }
        }
        for (int j = 0; j < 4; ++j) {
            const int i = col + j;
            if (i >= a.I) {
                float e = 0.f;
                if (i < a.I + a.E) {
                    const int eo = i - a.I;
                    e = a.emb_b[eo];
                    for (int f = 0; f < a.E; ++f) e += a.emb_w[eo * a.E + f] * (has_emb ? s_temb[f] : temb_value((float)t, f, a.E));
                    if (a.temb_out) a.temb_out[(int64_t)b * a.E + eo] = has_emb ? s_temb[eo] : temb_value((float)t, eo, a.E);
                }
                v[j] = e;
            }
        }
        *reinterpret_cast<f32x4*>(a.xin + (int64_t)b * a.ldxin + col) = f32x4{v[0], v[1], v[2], v[3]};
        if (a.xin16 && col < a.ldxin16) {
            const uint2 w = make_uint2(gd_bf16_bits(v[0]) | ((unsigned)gd_bf16_bits(v[1]) << 16),
                                       gd_bf16_bits(v[2]) | ((unsigned)gd_bf16_bits(v[3]) << 16));
            *reinterpret_cast<uint2*>(a.xin16 + (int64_t)b * a.ldxin16 + col) = w;
        }
        // column I + E of the float32 matrix (the first padding column, when there is one) holds 1: as one more column of the first
        // layer's weight-gradient product's operand it makes that layer's bias gradient a column of the product
        // (gdmcf_linear_bwd_weight_f32, a_scale_col).  The bf16 shadow keeps its zero there.  (Same thread, same address, program order.)
        const int oc = a.I + a.E;
        if (oc < a.ldxin && col <= oc && oc < col + 4) a.xin[(int64_t)b * a.ldxin + oc] = 1.f;
    }
}

// One-hot rows with discrete transition noise (reference gaussian_diffusion.py:770-831 with :597-614, :999-1038, and the
// `x_tU & one_hot(x_start)` of :849 / :686): item i of row b has class c0 = x0[b,i]; a class s is drawn from row c0 of
// Q = a*I + (1-a)*[[e,1-e],[e,1-e]], a = (float)ts[b] / B (the reference's own scaling, :775); the pair written is
// (c0==0 && s==0, c0==1 && s==1), i.e. the true class's bit survives only where the draw reproduces it.
// Four items per thread: one Philox block (stream 3) gives their four uniforms.
__global__ __launch_bounds__(256) void onehot_noise_kernel(const float* __restrict__ x0, int64_t ldx,
                                                          const int64_t* __restrict__ ts, int B, int I, float p1_off,
                                                          const uint8_t* __restrict__ sampled, int64_t lds, uint64_t seed,
                                                          uint64_t offset, float* __restrict__ xU, int64_t ldu,
                                                          uint8_t* __restrict__ sampled_out, int64_t ldso) {
    const int b = blockIdx.y;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= I) return;
    uint32_t u[4] = {0u, 0u, 0u, 0u};
    float a = 1.f;
    if (!sampled) {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)(i0 >> 2), (uint32_t)b, 3u, (uint32_t)offset),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        u[0] = r.x; u[1] = r.y; u[2] = r.z; u[3] = r.w;
        a = __fdiv_rn((float)ts[b], (float)B);
    }
    typedef f32x4 f32x4_u4 __attribute__((aligned(4)));
    const bool full = i0 + 3 < I;
    float xv[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t sv = 0;  // given classes of the four items, one per byte
    if (full) {  // rows are only 4-byte aligned when I is odd: gfx950 takes unaligned 16-byte accesses
        const f32x4 t4 = *reinterpret_cast<const f32x4_u4*>(x0 + (int64_t)b * ldx + i0);
        xv[0] = t4.x; xv[1] = t4.y; xv[2] = t4.z; xv[3] = t4.w;
    } else {
        for (int j = 0; j < 4; ++j)
            if (i0 + j < I) xv[j] = x0[(int64_t)b * ldx + i0 + j];
    }
    if (sampled)
        for (int j = 0; j < 4; ++j)
            if (i0 + j < I) sv |= (uint32_t)(sampled[(int64_t)b * lds + i0 + j] != 0) << (8 * j);
    float o[8];
    uint32_t so = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c0 = xv[j] != 0.f;
        int s;
        if (sampled) {
            s = (sv >> (8 * j)) & 1;
        } else {
            // P(class 1) = a*[c0 == 1] + (1 - a)*(1 - e), each product and the sum rounded to f32 as torch does
            float p1;
            {
#pragma clang fp contract(off)
                const float q = (1.f - a) * p1_off;
                p1 = (c0 ? a : 0.f) + q;
            }
            s = ((float)(u[j] >> 8) * 5.9604644775390625e-8f) < p1;
        }
        so |= (uint32_t)s << (8 * j);
        const float keep = (s == c0) ? 1.f : 0.f;
        o[2 * j] = c0 ? 0.f : keep;
        o[2 * j + 1] = c0 ? keep : 0.f;
    }
    float* op = xU + (int64_t)b * ldu + 2 * (int64_t)i0;
    if (full) {
        *reinterpret_cast<f32x4_u4*>(op) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4_u4*>(op + 4) = f32x4{o[4], o[5], o[6], o[7]};
    } else {
        for (int j = 0; j < 4; ++j)
            if (i0 + j < I) {
                op[2 * j] = o[2 * j];
                op[2 * j + 1] = o[2 * j + 1];
            }
    }
    if (sampled_out)
        for (int j = 0; j < 4; ++j)
            if (i0 + j < I) sampled_out[(int64_t)b * ldso + i0 + j] = (uint8_t)((so >> (8 * j)) & 1);
}

// ---------------------------------------------------------------------------------------------
// N(0,1) fill (reference gaussian_diffusion.py:328-331 `noise = th.randn_like(x_start)` when eps is the TARGET and so has to
// exist in memory, :210-217 the reverse loop's `noise = th.randn_like(x_t)`): the SAME normals the input builder draws in
// place (noise_mode 2) for the same (seed, offset) when stream == 0 -- element (b, i) is normal i & 3 of the block with
// counter (i >> 2, b, stream, offset), Box-Muller on (x, y) and (z, w) -- so a row written here and handed to the builder
// as given noise reproduces the in-kernel stream bit for bit.  Four elements per thread, 16-byte stores where the row allows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, int64_t ld, int rows, int cols, uint32_t stream,
                                                   uint64_t seed, uint64_t offset) {
    const int b = blockIdx.y;
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    float* __restrict__ orow = out + (int64_t)b * ld;
    typedef f32x4 f32x4_u4 __attribute__((aligned(4)));
#pragma unroll
    for (int u = 0; u < PREP_G; ++u) {
        const int col = (blockIdx.x * (256 * PREP_G) + u * 256 + threadIdx.x) * 4;
        if (col >= cols) return;
        const uint4 r = philox4x32_10(make_uint4((uint32_t)(col >> 2), (uint32_t)b, stream, (uint32_t)offset), key);
        float z[4];
        box_muller(r.x, r.y, z[0], z[1]);
        box_muller(r.z, r.w, z[2], z[3]);
        if (col + 3 < cols) {
            *reinterpret_cast<f32x4_u4*>(orow + col) = f32x4{z[0], z[1], z[2], z[3]};
        } else {
            for (int j = 0; j < 4; ++j)
                if (col + j < cols) orow[col + j] = z[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Loss target of the eps parameterisation (reference gaussian_diffusion.py:344-348): target = eps, except rows with t == 0
// (when the x0-likelihood term is on) whose target is r1[0]*x_t - x0 with weight r2[0] on the model output and twice the
// divisor.  target may BE the noise buffer: then only the t == 0 rows are touched (a few KB instead of three [B, I] passes).
// The product and the difference are rounded separately, as torch's mul and sub are.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eps_target_kernel(const float* __restrict__ noise, int64_t ldn, const float* __restrict__ xt,
                                                        int64_t ldxt, const float* __restrict__ x0, int64_t ldx0,
                                                        const int64_t* __restrict__ ts, const float* __restrict__ r1,
                                                        const float* __restrict__ r2, int t0_likelihood, int I,
                                                        float* target, int64_t ldt, float* __restrict__ alpha,
                                                        float* __restrict__ rowdiv) {
    const int b = blockIdx.y;
    const bool is0 = t0_likelihood && ts[b] == 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        alpha[b] = is0 ? r2[0] : 1.f;
        rowdiv[b] = is0 ? 2.f * (float)I : (float)I;
    }
    if (!is0 && target == noise) return;
    const float c = r1[0];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < I; i += gridDim.x * 256) {
        float v;
        if (is0) {
#pragma clang fp contract(off)
            const float p = c * xt[(int64_t)b * ldxt + i];
            v = p - x0[(int64_t)b * ldx0 + i];
        } else {
            v = noise[(int64_t)b * ldn + i];
        }
        target[(int64_t)b * ldt + i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// degree-guided graph of the reverse loop (reference gaussian_diffusion.py:706-729): per reverse step the reference
// draws a class for every (user, item) from row c of Q_bar(t / batch) where c is the edge's state so far
// (apply_noise on the accumulated one-hot graph), draws ONE bit per user from [1 - deg/maxdeg, deg/maxdeg]
// (multinomial(1)), ANDs the two when args.user_guided and ORs the result into the graph.  As bits:
//   graph[b,i] |= s[b,i] & (user_guided ? pick[b] : 1),   s ~ (u < a*[c == 1] + (1 - a)*(1 - e)),  a = (float)ts[b]/B.
// One byte per edge state, four items per thread; Philox4x32-10 streams 5 (classes) and 6 (user bits).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void graph_step_kernel(uint8_t* __restrict__ graph, int64_t ldg, const int64_t* __restrict__ ts,
                                                        int B, int I, float p1_off, const uint8_t* __restrict__ sampled,
                                                        int64_t lds, const uint8_t* __restrict__ pick_in,
                                                        const float* __restrict__ degp, int user_guided, uint64_t seed,
                                                        uint64_t offset, uint8_t* __restrict__ sampled_out, int64_t ldso,
                                                        uint8_t* __restrict__ pick_out) {
    const int b = blockIdx.y;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    int pick = 1;
    if (pick_in) {
        pick = pick_in[b] != 0;
    } else if (degp) {  // one draw per user, the same in every thread of the row
        const uint4 r = philox4x32_10(make_uint4(0xFFFFFFFFu, (uint32_t)b, 6u, (uint32_t)offset), key);
        pick = ((float)(r.x >> 8) * 5.9604644775390625e-8f) < degp[b];
    }
    if (pick_out && blockIdx.x == 0 && threadIdx.x == 0) pick_out[b] = (uint8_t)pick;
    if (i0 >= I) return;
    uint32_t u[4] = {0u, 0u, 0u, 0u};
    float a = 1.f;
    if (!sampled) {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)(i0 >> 2), (uint32_t)b, 5u, (uint32_t)offset), key);
        u[0] = r.x; u[1] = r.y; u[2] = r.z; u[3] = r.w;
        a = __fdiv_rn((float)ts[b], (float)B);
    }
    const int gate = user_guided ? pick : 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (i0 + j >= I) break;
        uint8_t* gp = graph + (int64_t)b * ldg + i0 + j;
        const int c = *gp != 0;
        int s;
        if (sampled) {
            s = sampled[(int64_t)b * lds + i0 + j] != 0;
        } else {
            float p1;
            {
#pragma clang fp contract(off)
                const float q = (1.f - a) * p1_off;
                p1 = (c ? a : 0.f) + q;
            }
            s = ((float)(u[j] >> 8) * 5.9604644775390625e-8f) < p1;
        }
        if (sampled_out) sampled_out[(int64_t)b * ldso + i0 + j] = (uint8_t)s;
        *gp = (uint8_t)(c | (s & gate));
    }
}

// ---------------------------------------------------------------------------------------------
// pieces of the indexIn backbone (reference models/DNN.py:510-682): embedding-row gather / scatter, row norms and
// the backward of x / |x| for the cosine scores, tanh' on a gradient with an extra addend.  All HBM-bound, one
// workgroup per row (16-byte accesses along the row), reductions in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ X, int64_t ld, int cols,
                                                       float* __restrict__ norm, float* __restrict__ inv_norm) {
    typedef f32x4 f32x4_u4 __attribute__((aligned(4)));
    __shared__ float red[4];
    const float* x = X + (int64_t)blockIdx.x * ld;
    float ss = 0.f;
    const int c4 = cols & ~3;
    for (int c = threadIdx.x * 4; c < c4; c += 1024) {
        const f32x4 t = *reinterpret_cast<const f32x4_u4*>(x + c);
        ss += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
    }
    if (threadIdx.x < cols - c4) ss += x[c4 + threadIdx.x] * x[c4 + threadIdx.x];
    const float n = sqrtf(block_sum_256(ss, red));
    if (threadIdx.x == 0) {
        if (norm) norm[blockIdx.x] = n;
        if (inv_norm) inv_norm[blockIdx.x] = 1.f / n;
    }
}

// dX = (dY - Y * <dY, Y>) * inv_norm  for Y = X / |X| (row-wise); dX may alias dY.  Rows of up to 4096 columns stay in
// registers between the dot product and the update (each operand is read once); longer rows are read twice.
__global__ __launch_bounds__(256) void normalize_rows_bwd_kernel(const float* __restrict__ dY, int64_t lddy,
                                                                const float* __restrict__ Y, int64_t ldy,
                                                                const float* __restrict__ inv_norm, int cols,
                                                                float* __restrict__ dX, int64_t lddx) {
    __shared__ float red[4];
    const float* dy = dY + (int64_t)blockIdx.x * lddy;
    const float* y = Y + (int64_t)blockIdx.x * ldy;
    float* dx = dX + (int64_t)blockIdx.x * lddx;
    const float rn = inv_norm[blockIdx.x];
    if (cols <= 4096 && (cols & 3) == 0) {
        typedef f32x4 f32x4_u4 __attribute__((aligned(4)));
        f32x4 a[4], b[4];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (threadIdx.x + 256 * k) * 4;
            a[k] = b[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < cols) {
                a[k] = *reinterpret_cast<const f32x4_u4*>(dy + c);
                b[k] = *reinterpret_cast<const f32x4_u4*>(y + c);
            }
            dot += a[k].x * b[k].x + a[k].y * b[k].y + a[k].z * b[k].z + a[k].w * b[k].w;
        }
        dot = block_sum_256(dot, red);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (threadIdx.x + 256 * k) * 4;
            if (c < cols) *reinterpret_cast<f32x4_u4*>(dx + c) = (a[k] - b[k] * dot) * rn;
        }
        return;
    }
    float dot = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) dot += dy[c] * y[c];
    dot = block_sum_256(dot, red);
    for (int c = threadIdx.x; c < cols; c += 256) dx[c] = (dy[c] - y[c] * dot) * rn;
}

__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dA, int64_t ldd, const float* __restrict__ A,
                                                      int64_t lda, const float* __restrict__ extra, int64_t lde,
                                                      const float* __restrict__ scale, int N, float* __restrict__ out,
                                                      int64_t ldo) {
    const int m = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float g = dA[(int64_t)m * ldd + n];
    if (extra) g += scale[0] * extra[(int64_t)m * lde + n];
    const float a = A[(int64_t)m * lda + n];
    out[(int64_t)m * ldo + n] = g * (1.f - a * a);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t lds,
                                                         const int64_t* __restrict__ index, int cols,
                                                         float* __restrict__ dst, int64_t ldd) {
    const float* s = src + index[blockIdx.x] * lds;
    float* d = dst + (int64_t)blockIdx.x * ldd;
    for (int c = threadIdx.x; c < cols; c += 256) d[c] = s[c];
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, int64_t lds,
                                                              const int64_t* __restrict__ index, int cols,
                                                              float* __restrict__ dst, int64_t ldd) {
    const float* s = src + (int64_t)blockIdx.x * lds;
    float* d = dst + index[blockIdx.x] * ldd;
    for (int c = threadIdx.x; c < cols; c += 256) atomicAdd(d + c, s[c]);  // (rows of one batch are distinct users)
}

__global__ void emb_cols_kernel(const int64_t* __restrict__ ts, const float* __restrict__ emb_w,
                                const float* __restrict__ emb_b, int E, int I, float* __restrict__ xin, int64_t ldxin,
                                float* __restrict__ temb_out, unsigned short* __restrict__ xin16, int64_t ldxin16) {
    const int b = blockIdx.x;
    const float t = (float)ts[b];
    // the E sinusoids once per row, one per lane (not E + 1 of them in a serial chain of libm calls per embedding column)
    __shared__ float s_temb[256];
    const bool staged = E <= 256;
    if (staged) {
        for (int f = threadIdx.x; f < E; f += blockDim.x) s_temb[f] = temb_value(t, f, E);
        __syncthreads();
    }
    for (int i = I + threadIdx.x; i < ldxin; i += blockDim.x) {
        float e = 0.f;
        if (i < I + E) {
            const int eo = i - I;
            e = emb_b[eo];
            for (int f = 0; f < E; ++f) e += emb_w[eo * E + f] * (staged ? s_temb[f] : temb_value(t, f, E));
            if (temb_out) temb_out[(int64_t)b * E + eo] = staged ? s_temb[eo] : temb_value(t, eo, E);
        }
        xin[(int64_t)b * ldxin + i] = e;
        if (xin16 && i < ldxin16) xin16[(int64_t)b * ldxin16 + i] = gd_bf16_bits(e);
    }
}

// ---------------------------------------------------------------------------------------------
// split-K slab reducers (fixed slab order -> deterministic)
// ---------------------------------------------------------------------------------------------
// mode 0: out = act(sum + bias[n]);  mode 1: out = rowscale[m] * sum * (act ? 1 - aact^2 : 1)
// Four consecutive columns per thread (one 16-byte load per slab; the slab rows are 16-byte aligned, see
// gdmcf_linear_ws_bytes) -- the additions per element are in the same slab order as before.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int64_t slab_stride,
                                                            int splits, int64_t ld_slab, int M, int N, int mode,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ rowscale,
                                                            const float* __restrict__ aact, int64_t ldact, int act,
                                                            float* __restrict__ out, int64_t ldo,
                                                            unsigned short* __restrict__ out16, int64_t ldo16, int vec) {
    const int m = blockIdx.y;
    const int n0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (n0 >= N) return;
    float sv[4] = {0.f, 0.f, 0.f, 0.f};
    const float* p = slabs + (int64_t)m * ld_slab + n0;
    if (vec && n0 + 3 < N) {
        // eight slab loads in flight, added in slab order (a plain loop waits for every load before issuing the next:
        // 19 x the L2 latency was the whole 9 us of this kernel)
        for (int k0 = 0; k0 < splits; k0 += 8) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                t[u] = (k0 + u < splits) ? *reinterpret_cast<const f32x4*>(p + (int64_t)(k0 + u) * slab_stride)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + u < splits) { sv[0] += t[u].x; sv[1] += t[u].y; sv[2] += t[u].z; sv[3] += t[u].w; }
        }
    } else {
        for (int k = 0; k < splits; ++k)
            for (int j = 0; j < 4; ++j)
                if (n0 + j < N) sv[j] += p[(int64_t)k * slab_stride + j];
    }
    const float rs = (mode != 0 && rowscale) ? rowscale[m] : 1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j;
        if (n >= N) break;
        float s = sv[j];
        if (mode == 0) {
            if (bias) s += bias[n];
            if (act == 1) s = tanhf(s);
        } else {
            if (rowscale) s *= rs;
            if (act == 1) {
                const float h = aact[(int64_t)m * ldact + n];
                s *= (1.f - h * h);
            }
        }
        out[(int64_t)m * ldo + n] = s;
        if (out16) out16[(int64_t)m * ldo16 + n] = gd_bf16_bits(s);
    }
}

// out[m, k] = A[m, k] * rs[m]; four columns per thread (16-byte accesses; rows need only 4-byte alignment on gfx950)
__global__ __launch_bounds__(256) void rowscale_kernel(const float* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ rs, int M, int K,
                                                       float* __restrict__ out, int64_t ldo,
                                                       unsigned short* __restrict__ out16, int64_t ldo16, int bias_col) {
    typedef f32x4 f32x4_u4 __attribute__((aligned(4)));
    const int m = blockIdx.y, k = (blockIdx.x * 256 + threadIdx.x) * 4;
    // column K of the scaled copy = the row scale itself: as one more column of the weight-gradient product's second operand it
    // makes the bias gradient sum_m rs[m] dZ[m, n] column K of that product (gdmcf_linear_bwd_weight_f32)
    if (bias_col && blockIdx.x == 0 && threadIdx.x == 0) out[(int64_t)m * ldo + K] = rs[m];
    if (k >= K) return;
    const float r = rs[m];
    const float* a = A + (int64_t)m * lda + k;
    float* o = out + (int64_t)m * ldo + k;
    float v[4];
    if (k + 3 < K) {
        const f32x4 t = *reinterpret_cast<const f32x4_u4*>(a);
        v[0] = t.x * r; v[1] = t.y * r; v[2] = t.z * r; v[3] = t.w * r;
        *reinterpret_cast<f32x4_u4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
        for (int j = 0; j < 4; ++j)
            if (k + j < K) o[j] = v[j] = a[j] * r;
    }
    if (out16)
        for (int j = 0; j < 4; ++j)
            if (k + j < K) out16[(int64_t)m * ldo16 + k + j] = gd_bf16_bits(v[j]);
}

// db[n] = sum_m rs[m]*dZ[m,n].  One workgroup per 64 columns; wave w sums rows w, w+4, ... (each row read
// is one coalesced 256-B segment, 8 of them in flight), then the four partial sums are added in wave order.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dZ, int64_t ld,
                                                     const float* __restrict__ rs, int M, int N,
                                                     float* __restrict__ db) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int nc = min(n, N - 1);
    float s = 0.f;
    int m = wave;
    for (; m + 28 < M; m += 32) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = dZ[(int64_t)(m + 4 * j) * ld + nc] * (rs ? rs[m + 4 * j] : 1.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; m < M; m += 4) s += dZ[(int64_t)m * ld + nc] * (rs ? rs[m] : 1.f);
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && n < N) db[n] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// W1e[n, e] = W1[n, I+e]: the E embedding columns of the first layer gathered into a compact [N, E] block
__global__ __launch_bounds__(256) void emb_gather_w_kernel(const float* __restrict__ W1, int64_t ldw, int I, int E, int N,
                                                           float* __restrict__ W1e) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < N * E) W1e[idx] = W1[(int64_t)(idx / E) * ldw + I + (idx % E)];
}

// demb[m,e] = sum_n dZ1[m,n] * W1e[n,e]   (one workgroup per row m, waves stride over e)
__global__ __launch_bounds__(256) void emb_bwd_demb_kernel(const float* __restrict__ dZ1, int64_t lddz,
                                                           const float* __restrict__ W1e, int E, int N,
                                                           float* __restrict__ demb) {
    const int m = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = wave; e < E; e += 4) {
        float s = 0.f;
        for (int n = lane; n < N; n += 64) s += dZ1[(int64_t)m * lddz + n] * W1e[(int64_t)n * E + e];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) demb[(int64_t)m * E + e] = s;
    }
}

// dWe[e,f] = sum_m demb[m,e]*temb[m,f];  dbe[e] = sum_m demb[m,e]   (one wave per output element)
__global__ __launch_bounds__(64) void emb_bwd_w_kernel(const float* __restrict__ demb, const float* __restrict__ temb,
                                                       int M, int E, float* __restrict__ dWe,
                                                       float* __restrict__ dbe) {
    const int idx = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    if (idx < E * E) {
        const int e = idx / E, f = idx % E;
        for (int m = lane; m < M; m += 64) s += demb[(int64_t)m * E + e] * temb[(int64_t)m * E + f];
    } else {
        const int e = idx - E * E;
        for (int m = lane; m < M; m += 64) s += demb[(int64_t)m * E + e];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        if (idx < E * E) dWe[idx] = s;
        else dbe[idx - E * E] = s;
    }
}

// rowsum[m] = sum_j rowpart[m, j]: one wave per row, lane-strided partials + xor tree (fixed order)
__global__ __launch_bounds__(256) void rowpart_reduce_kernel(const float* __restrict__ rowpart, int ld, int M, int nt,
                                                             float* __restrict__ rowsum) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    float s = 0.f;
    for (int j = lane; j < nt; j += 64) s += rowpart[(int64_t)m * ld + j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) rowsum[m] = s;
}

// ---------------------------------------------------------------------------------------------
// float64 loss tail + Lt-history FIFO (reference gaussian_diffusion.py:339-370)
// ---------------------------------------------------------------------------------------------
// The reference appends row by row (FIFO of H per timestep, :355-368).  The end state only depends,
// per timestep t, on the order of the rows with ts == t: it is the last min(H, cnt+n_t) entries of
// [old entries..., new entries in batch order].  So every row computes its rank among the earlier
// rows with the same t (independent, pipelined LDS reads -- no serial dependency chain) and writes
// straight to its final slot.  Needs T*H doubles + B ints of LDS.
__device__ void lt_history_parallel(const int64_t* __restrict__ ts, const double* __restrict__ lu, int B, int T, int H,
                                    double* hist, int64_t* cnt, unsigned char* lds_raw) {
    double* old = reinterpret_cast<double*>(lds_raw);           // [T*H]
    int* n_t = reinterpret_cast<int*>(old + (size_t)T * H);     // [T]
    int* c0 = n_t + T;                                          // [T]
    int* tsl = c0 + T;                                          // [B]
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int i = tid; i < T * H; i += nth) old[i] = hist[i];
    for (int t = tid; t < T; t += nth) {
        n_t[t] = 0;
        c0[t] = (int)cnt[t];
    }
    for (int b = tid; b < B; b += nth) tsl[b] = (int)ts[b];
    __syncthreads();
    for (int b = tid; b < B; b += nth) atomicAdd(&n_t[tsl[b]], 1);
    __syncthreads();
    // old entries slide left by `drop`
    for (int i = tid; i < T * H; i += nth) {
        const int t = i / H, j = i % H;
        const int drop = max(0, c0[t] + n_t[t] - H);
        if (j < c0[t] && j - drop >= 0) hist[(int64_t)t * H + (j - drop)] = old[i];
    }
    // new entries
    for (int b = tid; b < B; b += nth) {
        const int t = tsl[b];
        int rank = 0;
        for (int p = 0; p < b; ++p) rank += (tsl[p] == t);
        const int drop = max(0, c0[t] + n_t[t] - H);
        const int pos = c0[t] + rank - drop;
        if (pos >= 0) hist[(int64_t)t * H + pos] = lu[b];
    }
    for (int t = tid; t < T; t += nth) cnt[t] = (int64_t)min(H, c0[t] + n_t[t]);
}

__global__ __launch_bounds__(256) void row_loss_finish_kernel(const float* __restrict__ rowsum,
                                                              const float* __restrict__ rowdiv,
                                                              const float* __restrict__ alpha,
                                                              float* __restrict__ gradcoef,
                                                              const int64_t* __restrict__ ts,
                                                              const double* __restrict__ weight_t,
                                                              const double* __restrict__ pt, int B, int T, int H,
                                                              double* hist, int64_t* cnt, int update,
                                                              double* __restrict__ lu, double* __restrict__ loss,
                                                              double* __restrict__ loss_mean, float* __restrict__ rowscale_mean,
                                                              float inv_b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fin_lds[];
    double part = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float mse = rowsum[b] / rowdiv[b];  // f32 mean, as mean_flat on f32 (:335)
        const double l = weight_t[ts[b]] * (double)mse;  // f64 weight * f32 mse -> f64 (:352)
        lu[b] = l;
        const double lb = l / pt[b];  // (:370)
        loss[b] = lb;
        part += lb;
        if (gradcoef) {
            const float gc = (float)(2.0 * (alpha ? (double)alpha[b] : 1.0) * weight_t[ts[b]] / (pt[b] * (double)rowdiv[b]));
            gradcoef[b] = gc;
            // the mean reduction of main.py:348 has the constant upstream gradient 1/B: the row scale of its backward
            if (rowscale_mean) rowscale_mean[b] = gc * inv_b;
        }
    }
    if (loss_mean) {  // losses["loss"].mean() (main.py:348) in float64, fixed order: lane-strided sums, xor tree, waves 0..3
        double* s_part = reinterpret_cast<double*>(fin_lds);  // the FIFO update below re-initialises what it uses
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) *loss_mean = (((s_part[0] + s_part[1]) + s_part[2]) + s_part[3]) / (double)B;
    }
    __syncthreads();
    if (update) lt_history_parallel(ts, lu, B, T, H, hist, cnt, fin_lds);
}

__global__ __launch_bounds__(256) void lt_history_update_kernel(const int64_t* ts, const double* lu, int B, int T,
                                                                int H, double* hist, int64_t* cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fin_lds[];
    lt_history_parallel(ts, lu, B, T, H, hist, cnt, fin_lds);
}

// ---------------------------------------------------------------------------------------------
// importance-sampled timesteps (reference gaussian_diffusion.py:373-397) in ONE launch, no host sync:
// uniform until every Lt_count == H, then p_t ~ sqrt(mean(Lt_history^2)) mixed with uniform_prob,
// inverse-CDF sampling with a Philox stream; pt = p[t]*T (float64), or 1 in the uniform phase.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_timesteps_kernel(const double* __restrict__ hist,
                                                               const int64_t* __restrict__ cnt, int T, int H, int B,
                                                               double uniform_prob, uint64_t seed, uint64_t offset,
                                                               int64_t* __restrict__ ts, double* __restrict__ pt,
                                                               double* __restrict__ p_out, const GdStepState* step_state) {
    if (step_state) offset = step_state->ts_offset;
    extern __shared__ __attribute__((aligned(16))) unsigned char st_lds[];
    double* p = reinterpret_cast<double*>(st_lds);  // [T] probabilities, then inclusive CDF in cdf[]
    double* cdf = p + T;
    __shared__ int full;
    __shared__ double total;
    const int tid = threadIdx.x;
    if (tid == 0) full = 1;
    __syncthreads();
    for (int t = tid; t < T; t += 256)
        if (cnt[t] != H) full = 0;
    __syncthreads();
    const bool imp = (full != 0);
    if (imp) {
        for (int t = tid; t < T; t += 256) {
            double s = 0.0;
            for (int j = 0; j < H; ++j) {
                const double v = hist[(int64_t)t * H + j];
                s += v * v;
            }
            p[t] = sqrt(s / (double)H);
        }
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int t = 0; t < T; ++t) s += p[t];
            total = s;
        }
        __syncthreads();
        for (int t = tid; t < T; t += 256) {
            double v = p[t] / total;
            v *= 1.0 - uniform_prob;
            v += uniform_prob / (double)T;
            p[t] = v;
            if (p_out) p_out[t] = v;
        }
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int t = 0; t < T; ++t) {
                s += p[t];
                cdf[t] = s;
            }
        }
        __syncthreads();
    }
    for (int b = tid; b < B; b += 256) {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)b, 0u, 2u, (uint32_t)offset),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        const double u = ((double)r.x * 4294967296.0 + (double)r.y) * (1.0 / 18446744073709551616.0);  // [0,1)
        int t;
        if (imp) {
            const double x = u * cdf[T - 1];
            int lo = 0, hi = T - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] > x) hi = mid; else lo = mid + 1;
            }
            t = lo;
            pt[b] = p[t] * (double)T;
        } else {
            t = min((int)(u * (double)T), T - 1);
            pt[b] = 1.0;
        }
        ts[b] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// fused multi-tensor AdamW (torch.optim.AdamW single-tensor math, reference main.py:258,351)
// ---------------------------------------------------------------------------------------------
constexpr int ADAM_BLOCK_ELEMS = 4096;

typedef GdAdamHyper AdamHyper;
#define adam_elem gd_adam_elem

// stab (optional): [n][3] = (bf16 shadow pointer or 0, columns, shadow row stride) -- the updated parameter is also
// stored, rounded to bfloat16, into its zero-padded 2-D shadow (gdmcf_bf16_shadow_set)
template <bool NT_>
__global__ __launch_bounds__(256) void adamw_kernel(const int64_t* __restrict__ table, int n_tensors,
                                                    AdamHyper h, const int64_t* __restrict__ stab, const GdStepState* step_state) {
    if (step_state) h = step_state->hyper;  // graph mode: this step's scalars from the device
    int t = 0;
    for (int i = 1; i < n_tensors; ++i)
        if ((int64_t)blockIdx.x >= table[i * 6 + 5]) t = i;
    float* p = reinterpret_cast<float*>(table[t * 6 + 0]);
    const float* g = reinterpret_cast<const float*>(table[t * 6 + 1]);
    float* m = reinterpret_cast<float*>(table[t * 6 + 2]);
    float* v = reinterpret_cast<float*>(table[t * 6 + 3]);
    const int64_t n = table[t * 6 + 4];
    const int64_t base = ((int64_t)blockIdx.x - table[t * 6 + 5]) * ADAM_BLOCK_ELEMS;
    const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                      reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
    unsigned short* p16 = stab ? reinterpret_cast<unsigned short*>(stab[t * 3 + 0]) : nullptr;
    const unsigned cols16 = p16 ? (unsigned)stab[t * 3 + 1] : 1u;
    const int64_t ld16 = p16 ? stab[t * 3 + 2] : 0;
#pragma unroll
    for (int it = 0; it < ADAM_BLOCK_ELEMS / (256 * 4); ++it) {
        const int64_t i = base + (int64_t)(it * 256 + threadIdx.x) * 4;
        if (al && i + 3 < n) {
            f32x4 pp, gg, mm, vv;
            if (NT_) {
                pp = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p + i));
                gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i));
                mm = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m + i));
                vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v + i));
            } else {
                pp = *reinterpret_cast<f32x4*>(p + i);
                gg = *reinterpret_cast<const f32x4*>(g + i);
                mm = *reinterpret_cast<f32x4*>(m + i);
                vv = *reinterpret_cast<f32x4*>(v + i);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pj = pp[j], mj = mm[j], vj = vv[j];
                adam_elem(pj, gg[j], mj, vj, h);
                pp[j] = pj;
                mm[j] = mj;
                vv[j] = vj;
            }
            if (NT_) {
                __builtin_nontemporal_store(pp, reinterpret_cast<f32x4*>(p + i));
                __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m + i));
                __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + i));
            } else {
                *reinterpret_cast<f32x4*>(p + i) = pp;
                *reinterpret_cast<f32x4*>(m + i) = mm;
                *reinterpret_cast<f32x4*>(v + i) = vv;
            }
            if (p16) {
                unsigned r = (unsigned)i / cols16, c = (unsigned)i - r * cols16;  // numel < 2^32 (checked on the host)
                if (c + 3 < cols16) {
                    // one 8-byte store; only 2-byte aligned when the row length is odd (gfx950 runs with unaligned
                    // global access enabled, hipcc emits global_store_dwordx2 for it)
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    typedef u32x2 u32x2_u __attribute__((aligned(2)));
                    const u32x2 w = {gd_bf16_bits(pp[0]) | ((unsigned)gd_bf16_bits(pp[1]) << 16),
                                     gd_bf16_bits(pp[2]) | ((unsigned)gd_bf16_bits(pp[3]) << 16)};
                    *reinterpret_cast<u32x2_u*>(p16 + (int64_t)r * ld16 + c) = w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        p16[(int64_t)r * ld16 + c] = gd_bf16_bits(pp[j]);
                        if (++c == cols16) {
                            c = 0;
                            ++r;
                        }
                    }
                }
            }
        } else {
            for (int j = 0; j < 4; ++j)
                if (i + j < n) {
                    adam_elem(p[i + j], g[i + j], m[i + j], v[i + j], h);
                    if (p16) {
                        const unsigned r = (unsigned)(i + j) / cols16, c = (unsigned)(i + j) - r * cols16;
                        p16[(int64_t)r * ld16 + c] = gd_bf16_bits(p[i + j]);
                    }
                }
        }
    }
}

// dense[b, :] = row row_ids[b] of a CSR matrix (values NULL -> 1.0).  One workgroup per row: 16-byte zero fill,
// barrier, scatter of the row's nonzeros.  Replaces scipy .todense() + a 55 MB host-to-device copy per batch.
__global__ __launch_bounds__(256) void densify_rows_kernel(const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ indices,
                                                           const float* __restrict__ values,
                                                           const int64_t* __restrict__ row_ids, int I,
                                                           float* __restrict__ out, int64_t ldo) {
    const int b = blockIdx.x;
    float* row = out + (int64_t)b * ldo;
    const bool al = ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
    if (al) {
        for (int i = threadIdx.x * 4; i + 3 < I; i += 1024) *reinterpret_cast<f32x4*>(row + i) = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = (I & ~3) + threadIdx.x; i < I; i += 256) row[i] = 0.f;
    } else {
        for (int i = threadIdx.x; i < I; i += 256) row[i] = 0.f;
    }
    __syncthreads();
    const int64_t u = row_ids ? row_ids[b] : b;
    for (int64_t j = indptr[u] + threadIdx.x; j < indptr[u + 1]; j += 256) {
        const int c = indices[j];
        if (c >= 0 && c < I) row[c] = values ? values[j] : 1.f;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ a, int64_t n, float s,
                                                    float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] * s;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int gdmcf_dnn_prep_input_f32(const float* x, int64_t ldx, const int64_t* ts, const float* ca, const float* cb,
                             int noise_mode, const float* noise, int64_t ldn, int drop_mode, const uint8_t* keep,
                             int64_t ldkeep, float drop_p, uint64_t seed, uint64_t offset, int normalize,
                             const float* emb_w, const float* emb_b, int E, int B, int I, float* xin, int64_t ldxin,
                             float* xt_out, int64_t ldxt, float* temb_out, float* rownorm_ws, void* stream) {
    GD_CHECK_SHAPE(B > 0 && I > 0 && E >= 0, "prep_input: empty batch");
    GD_CHECK_SHAPE(ldxin >= I + E && (ldxin % 4) == 0 && gd_aligned16(xin), "prep_input: xin must be 16B aligned, ld%4==0");
    GD_CHECK_SHAPE(ldx >= I, "prep_input: ldx < I");
    GD_CHECK_ARG((ca == nullptr) == (cb == nullptr), "prep_input: ca/cb must both be set or both NULL");
    GD_CHECK_ARG(noise_mode >= 0 && noise_mode <= 2 && drop_mode >= 0 && drop_mode <= 2, "prep_input: bad mode");
    GD_CHECK_ARG(noise_mode != 1 || (noise && ldn >= I), "prep_input: explicit noise missing");
    GD_CHECK_ARG(drop_mode != 1 || (keep && ldkeep >= I), "prep_input: explicit keep-mask missing");
    GD_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "prep_input: dropout p out of range");
    GD_CHECK_ARG(E == 0 || (emb_w && emb_b && ts), "prep_input: embedding weights / ts missing");
    GD_CHECK_ARG(!ca || ts, "prep_input: ts missing");
    PrepArgs a;
    a.x = x; a.ldx = ldx; a.ts = ts; a.ca = ca; a.cb = cb; a.noise_mode = ca ? noise_mode : 0; a.noise = noise;
    a.ldn = ldn; a.drop_mode = drop_mode; a.keep = keep; a.ldkeep = ldkeep; a.drop_scale = 1.0f / (1.0f - drop_p);
    a.keep_thresh = (uint32_t)fmin(fmax(rint((1.0 - (double)drop_p) * 65536.0), 0.0), 65536.0);
    a.seed = seed; a.offset = offset; a.step_state = t_gd_step_state; a.rownorm = nullptr; a.emb_w = emb_w; a.emb_b = emb_b; a.E = E; a.B = B;
    a.I = I; a.xin = xin; a.ldxin = ldxin; a.xt_out = xt_out; a.ldxt = ldxt; a.temb_out = temb_out;
    a.xin16 = nullptr; a.ldxin16 = 0;
    a.csr_indptr = nullptr; a.csr_indices = nullptr; a.csr_rows = nullptr; a.bits_out = nullptr; a.ldbits = 0;
    GdShadow sh;
    if (gd_shadow_lookup(xin, &sh) && sh.rows == B && sh.cols == I + E) {  // keep the bf16 shadow of xin in sync
        a.xin16 = (unsigned short*)sh.p16;
        a.ldxin16 = sh.ld16;
    }
    hipStream_t s = (hipStream_t)stream;
    if (normalize) {
        // F.normalize (reference models/DNN.py:75-76) needs the L2 norm of the noised row first
        GD_CHECK_ARG(rownorm_ws != nullptr, "prep_input: normalize needs rownorm_ws [B]");
        hipLaunchKernelGGL(prep_rowss_kernel, dim3(B), dim3(256), 0, s, a, rownorm_ws);
        a.rownorm = rownorm_ws;
    }
    dim3 grid(gd_cdiv((int)(ldxin / 4), 256 * PREP_G), B);
    {
        // algorithmic bytes: read x (+ explicit noise / keep-mask), write xin
        const double bytes = (double)B * I * (4.0 + (a.noise_mode == 1 ? 4.0 : 0.0) + (drop_mode == 1 ? 1.0 : 0.0)) +
                             (double)B * ldxin * 4.0;
        GdProfScope prof(7, bytes, s);
        hipLaunchKernelGGL(prep_input_kernel, grid, dim3(256), 0, s, a);
    }
    return gd_launch_status("prep_input");
}

int gdmcf_dnn_prep_input_csr_f32(const int64_t* indptr, const int32_t* indices, const int64_t* rows, const int64_t* ts,
                                 const float* ca, const float* cb, int noise_mode, const float* noise, int64_t ldn,
                                 int drop_mode, const uint8_t* keep, int64_t ldkeep, float drop_p, uint64_t seed,
                                 uint64_t offset, const float* emb_w, const float* emb_b, int E, int B, int I, float* xin,
                                 int64_t ldxin, float* temb_out, uint32_t* bits_out, int64_t ldbits, void* stream) {
    GD_CHECK_SHAPE(B > 0 && I > 0 && E >= 0, "prep_input_csr: empty batch");
    GD_CHECK_SHAPE(ldxin >= I + E && (ldxin % 4) == 0 && gd_aligned16(xin), "prep_input_csr: xin must be 16B aligned, ld%4==0");
    GD_CHECK_ARG(indptr && indices && rows, "prep_input_csr: CSR arrays / row ids missing");
    GD_CHECK_ARG(!bits_out || ldbits >= (I + 31) / 32, "prep_input_csr: ldbits < ceil(I/32)");
    GD_CHECK_ARG((ca == nullptr) == (cb == nullptr), "prep_input_csr: ca/cb must both be set or both NULL");
    GD_CHECK_ARG(noise_mode >= 0 && noise_mode <= 2 && drop_mode >= 0 && drop_mode <= 2, "prep_input_csr: bad mode");
    GD_CHECK_ARG(noise_mode != 1 || (noise && ldn >= I), "prep_input_csr: explicit noise missing");
    GD_CHECK_ARG(drop_mode != 1 || (keep && ldkeep >= I), "prep_input_csr: explicit keep-mask missing");
    GD_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "prep_input_csr: dropout p out of range");
    GD_CHECK_ARG(E == 0 || (emb_w && emb_b && ts), "prep_input_csr: embedding weights / ts missing");
    GD_CHECK_ARG(!ca || ts, "prep_input_csr: ts missing");
    PrepArgs a;
    a.x = nullptr; a.ldx = 0; a.ts = ts; a.ca = ca; a.cb = cb; a.noise_mode = ca ? noise_mode : 0; a.noise = noise;
    a.ldn = ldn; a.drop_mode = drop_mode; a.keep = keep; a.ldkeep = ldkeep; a.drop_scale = 1.0f / (1.0f - drop_p);
    a.keep_thresh = (uint32_t)fmin(fmax(rint((1.0 - (double)drop_p) * 65536.0), 0.0), 65536.0);
    a.seed = seed; a.offset = offset; a.step_state = t_gd_step_state; a.rownorm = nullptr; a.emb_w = emb_w; a.emb_b = emb_b; a.E = E; a.B = B;
    a.I = I; a.xin = xin; a.ldxin = ldxin; a.xt_out = nullptr; a.ldxt = 0; a.temb_out = temb_out;
    a.xin16 = nullptr; a.ldxin16 = 0;
    a.csr_indptr = indptr; a.csr_indices = indices; a.csr_rows = rows; a.bits_out = bits_out; a.ldbits = ldbits;
    GdShadow sh;
    if (gd_shadow_lookup(xin, &sh) && sh.rows == B && sh.cols == I + E) {
        a.xin16 = (unsigned short*)sh.p16;
        a.ldxin16 = sh.ld16;
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(gd_cdiv((int)(ldxin / 4), 256 * PREP_G), B);
    {
        // algorithmic bytes: write xin (+ explicit noise / keep-mask); the rows themselves are a few hundred bytes of CSR
        const double bytes = (double)B * I * ((a.noise_mode == 1 ? 4.0 : 0.0) + (drop_mode == 1 ? 1.0 : 0.0)) +
                             (double)B * ldxin * 4.0;
        GdProfScope prof(7, bytes, s);
        hipLaunchKernelGGL(prep_input_kernel, grid, dim3(256), 0, s, a);
    }
    return gd_launch_status("prep_input_csr");
}

int gdmcf_dnn_emb_cols_f32(const int64_t* ts, const float* emb_w, const float* emb_b, int E, int B, int I, float* xin,
                           int64_t ldxin, float* temb_out, void* stream) {
    GD_CHECK_SHAPE(B > 0 && I > 0 && E > 0 && ldxin >= I + E, "emb_cols: bad shape");
    GdShadow sh;
    const bool has16 = gd_shadow_lookup(xin, &sh) && sh.rows == B && sh.cols == I + E;
    hipLaunchKernelGGL(emb_cols_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, ts, emb_w, emb_b, E, I, xin, ldxin,
                       temb_out, has16 ? (unsigned short*)sh.p16 : nullptr, has16 ? sh.ld16 : 0);
    return gd_launch_status("emb_cols");
}

int gdmcf_onehot_noise_f32(const float* x0, int64_t ldx, const int64_t* ts, int B, int I, float discrete,
                           const uint8_t* sampled, int64_t lds, uint64_t seed, uint64_t offset, float* xU, int64_t ldu,
                           uint8_t* sampled_out, int64_t ldso, void* stream) {
    GD_CHECK_SHAPE(B > 0 && I > 0 && ldx >= I && ldu >= 2 * (int64_t)I, "onehot_noise: bad shape");
    GD_CHECK_ARG(x0 && xU && (sampled ? lds >= I : ts != nullptr) && (!sampled_out || ldso >= I),
                 "onehot_noise: null pointer / bad leading dimension");
    // u_x = th.tensor([[e, 1 - e], ...]): 1 - e is formed in double and rounded to float32 once
    const float p1_off = (float)(1.0 - (double)discrete);
    {
        // algorithmic bytes: read x0 (+ the given classes), write the [B, 2I] image
        GdProfScope prof(10, (double)B * I * (4.0 + 8.0 + (sampled ? 1.0 : 0.0)), (hipStream_t)stream);
        hipLaunchKernelGGL(onehot_noise_kernel, dim3(gd_cdiv(I, 1024), B), dim3(256), 0, (hipStream_t)stream, x0, ldx, ts, B,
                           I, p1_off, sampled, lds, seed, offset, xU, ldu, sampled_out, ldso);
    }
    return gd_launch_status("onehot_noise");
}

int gdmcf_randn_f32(float* out, int64_t ld, int rows, int cols, int stream_id, uint64_t seed, uint64_t offset, void* stream) {
    GD_CHECK_SHAPE(rows > 0 && cols > 0 && ld >= cols, "randn: bad shape");
    GD_CHECK_ARG(out && stream_id >= 0 && stream_id < 256, "randn: null pointer / bad stream id");
    {
        GdProfScope prof(11, (double)rows * cols * 4.0, (hipStream_t)stream);  // algorithmic bytes: the store
        hipLaunchKernelGGL(randn_kernel, dim3(gd_cdiv(cols, 256 * PREP_G * 4), rows), dim3(256), 0, (hipStream_t)stream, out,
                           ld, rows, cols, (uint32_t)stream_id, seed, offset);
    }
    return gd_launch_status("randn");
}

int gdmcf_eps_target_f32(const float* noise, int64_t ldn, const float* xt, int64_t ldxt, const float* x0, int64_t ldx0,
                         const int64_t* ts, const float* r1, const float* r2, int t0_likelihood, int B, int I, float* target,
                         int64_t ldt, float* alpha, float* rowdiv, void* stream) {
    GD_CHECK_SHAPE(B > 0 && I > 0 && ldn >= I && ldxt >= I && ldx0 >= I && ldt >= I, "eps_target: bad shape");
    GD_CHECK_ARG(noise && xt && x0 && ts && r1 && r2 && target && alpha && rowdiv, "eps_target: null pointer");
    GD_CHECK_ARG(target != noise || ldt == ldn, "eps_target: in-place target needs the noise buffer's leading dimension");
    hipLaunchKernelGGL(eps_target_kernel, dim3(gd_cdiv(I, 2048), B), dim3(256), 0, (hipStream_t)stream, noise, ldn, xt, ldxt, x0,
                       ldx0, ts, r1, r2, t0_likelihood, I, target, ldt, alpha, rowdiv);
    return gd_launch_status("eps_target");
}

int gdmcf_graph_guided_step_u8(uint8_t* graph, int64_t ldg, const int64_t* ts, int B, int I, float discrete,
                               const uint8_t* sampled_in, int64_t lds, const uint8_t* pick_in, const float* degree_prob,
                               int user_guided, uint64_t seed, uint64_t offset, uint8_t* sampled_out, int64_t ldso,
                               uint8_t* pick_out, void* stream) {
    GD_CHECK_SHAPE(B > 0 && I > 0 && ldg >= I, "graph_guided_step: bad shape");
    GD_CHECK_ARG(graph && (sampled_in ? lds >= I : ts != nullptr) && (!sampled_out || ldso >= I),
                 "graph_guided_step: null pointer / bad leading dimension");
    GD_CHECK_ARG(!user_guided || pick_in || degree_prob, "graph_guided_step: user_guided needs pick_in or degree_prob");
    const float p1_off = (float)(1.0 - (double)discrete);
    hipLaunchKernelGGL(graph_step_kernel, dim3(gd_cdiv(I, 1024), B), dim3(256), 0, (hipStream_t)stream, graph, ldg, ts, B, I,
                       p1_off, sampled_in, lds, pick_in, degree_prob, user_guided, seed, offset, sampled_out, ldso, pick_out);
    return gd_launch_status("graph_guided_step");
}

int gdmcf_row_norms_f32(const float* X, int64_t ld, int rows, int cols, float* norm, float* inv_norm, void* stream) {
    GD_CHECK_SHAPE(rows > 0 && cols > 0 && ld >= cols, "row_norms: bad shape");
    GD_CHECK_ARG(X && (norm || inv_norm), "row_norms: null pointer");
    hipLaunchKernelGGL(row_norms_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, X, ld, cols, norm, inv_norm);
    return gd_launch_status("row_norms");
}

int gdmcf_normalize_rows_bwd_f32(const float* dY, int64_t lddy, const float* Y, int64_t ldy, const float* inv_norm, int rows,
                                 int cols, float* dX, int64_t lddx, void* stream) {
    GD_CHECK_SHAPE(rows > 0 && cols > 0 && lddy >= cols && ldy >= cols && lddx >= cols, "normalize_rows_bwd: bad shape");
    GD_CHECK_ARG(dY && Y && inv_norm && dX, "normalize_rows_bwd: null pointer");
    hipLaunchKernelGGL(normalize_rows_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dY, lddy, Y, ldy, inv_norm,
                       cols, dX, lddx);
    return gd_launch_status("normalize_rows_bwd");
}

int gdmcf_tanh_bwd_f32(const float* dA, int64_t ldd, const float* A, int64_t lda, const float* extra, int64_t lde,
                       const float* scale, int M, int N, float* out, int64_t ldo, void* stream) {
    GD_CHECK_SHAPE(M > 0 && N > 0 && ldd >= N && lda >= N && ldo >= N && (!extra || lde >= N), "tanh_bwd: bad shape");
    GD_CHECK_ARG(dA && A && out && (!extra || scale), "tanh_bwd: null pointer");
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(gd_cdiv(N, 256), M), dim3(256), 0, (hipStream_t)stream, dA, ldd, A, lda, extra, lde,
                       scale, N, out, ldo);
    return gd_launch_status("tanh_bwd");
}

int gdmcf_gather_rows_f32(const float* src, int64_t lds, const int64_t* index, int n, int cols, float* dst, int64_t ldd,
                          void* stream) {
    GD_CHECK_SHAPE(n > 0 && cols > 0 && lds >= cols && ldd >= cols, "gather_rows: bad shape");
    GD_CHECK_ARG(src && index && dst, "gather_rows: null pointer");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, src, lds, index, cols, dst, ldd);
    return gd_launch_status("gather_rows");
}

int gdmcf_scatter_add_rows_f32(const float* src, int64_t lds, const int64_t* index, int n, int cols, float* dst, int64_t ldd,
                               void* stream) {
    GD_CHECK_SHAPE(n > 0 && cols > 0 && lds >= cols && ldd >= cols, "scatter_add_rows: bad shape");
    GD_CHECK_ARG(src && index && dst, "scatter_add_rows: null pointer");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, src, lds, index, cols, dst, ldd);
    return gd_launch_status("scatter_add_rows");
}

int gdmcf_rowscale_f32(const float* A, int64_t lda, const float* rowscale, int M, int K, float* out, int64_t ldo,
                       void* stream) {
    GD_CHECK_SHAPE(M > 0 && K > 0 && lda >= K && ldo >= K, "rowscale: bad shape");
    const int64_t n = (int64_t)M * K;
    GdShadow sh;
    const bool has16 = gd_shadow_lookup(out, &sh) && sh.rows == M && sh.cols == K;
    (void)n;
    const int bias_col = ldo > K;  // room for one more column: out[m, K] = rowscale[m] (see the kernel)
    hipLaunchKernelGGL(rowscale_kernel, dim3(gd_cdiv(K, 1024), M), dim3(256), 0, (hipStream_t)stream, A, lda,
                       rowscale, M, K, out, ldo, has16 ? (unsigned short*)sh.p16 : nullptr, has16 ? sh.ld16 : 0, bias_col);
    return gd_launch_status("rowscale");
}

int gdmcf_emb_bwd_f32(const float* dZ1, int64_t lddz, const float* W1, int64_t ldw, int I, int E, const float* temb,
                      int M, int N, float* demb_ws, float* dWe, float* dbe, void* stream) {
    GD_CHECK_SHAPE(M > 0 && N > 0 && E > 0 && ldw >= I + E && lddz >= N, "emb_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    float* W1e = demb_ws + (size_t)M * E;  // demb_ws holds [M*E] demb followed by [N*E] gathered weights
    hipLaunchKernelGGL(emb_gather_w_kernel, dim3(gd_cdiv(N * E, 256)), dim3(256), 0, s, W1, ldw, I, E, N, W1e);
    hipLaunchKernelGGL(emb_bwd_demb_kernel, dim3(M), dim3(256), 0, s, dZ1, lddz, W1e, E, N, demb_ws);
    hipLaunchKernelGGL(emb_bwd_w_kernel, dim3(E * E + E), dim3(64), 0, s, demb_ws, temb, M, E, dWe, dbe);
    return gd_launch_status("emb_bwd");
}

int gdmcf_row_loss_finish_f64(const float* rowsum, const float* rowdiv, const float* alpha, const int64_t* ts,
                              const double* weight_t, const double* pt, int B, int T, int H, double* Lt_history,
                              int64_t* Lt_count, int update_history, double* loss_unscaled, double* loss,
                              float* gradcoef, void* stream) {
    return gdmcf_row_loss_finish_mean_f64(rowsum, rowdiv, alpha, ts, weight_t, pt, B, T, H, Lt_history, Lt_count, update_history,
                                          loss_unscaled, loss, gradcoef, nullptr, nullptr, stream);
}

int gdmcf_row_loss_finish_mean_f64(const float* rowsum, const float* rowdiv, const float* alpha, const int64_t* ts,
                                   const double* weight_t, const double* pt, int B, int T, int H, double* Lt_history,
                                   int64_t* Lt_count, int update_history, double* loss_unscaled, double* loss,
                                   float* gradcoef, double* loss_mean, float* rowscale_mean, void* stream) {
    GD_CHECK_SHAPE(B > 0 && T > 0 && H > 0, "row_loss_finish: bad shape");
    GD_CHECK_ARG(!rowscale_mean || gradcoef, "row_loss_finish: rowscale_mean needs gradcoef");
    const size_t lds = (size_t)T * H * 8 + (size_t)T * 8 + (size_t)B * 4 + 16;
    GD_CHECK_ARG(lds <= 150 * 1024, "row_loss_finish: T*H and B too large for the LDS-resident FIFO update (150 KiB)");
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {  // T = 1000 diffusion steps x 10 history entries need 88 KB
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(row_loss_finish_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            gdmcf_set_error("row_loss_finish: hipFuncSetAttribute failed");
            return GDMCF_E_HIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(row_loss_finish_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, rowsum, rowdiv, alpha,
                       gradcoef, ts, weight_t, pt, B, T, H, Lt_history, Lt_count, update_history, loss_unscaled, loss,
                       loss_mean, rowscale_mean, 1.0f / (float)B);
    return gd_launch_status("row_loss_finish");
}

int gdmcf_lt_history_update(const int64_t* ts, const double* loss_unscaled, int B, int T, int H, double* Lt_history,
                            int64_t* Lt_count, void* stream) {
    GD_CHECK_SHAPE(B > 0 && T > 0 && H > 0, "lt_history_update: bad shape");
    const size_t lds = (size_t)T * H * 8 + (size_t)T * 8 + (size_t)B * 4 + 16;
    GD_CHECK_ARG(lds <= 150 * 1024, "lt_history_update: T*H and B too large for the LDS-resident FIFO update (150 KiB)");
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lt_history_update_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            gdmcf_set_error("lt_history_update: hipFuncSetAttribute failed");
            return GDMCF_E_HIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(lt_history_update_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, ts, loss_unscaled, B, T,
                       H, Lt_history, Lt_count);
    return gd_launch_status("lt_history_update");
}

int gdmcf_sample_timesteps(const double* Lt_history, const int64_t* Lt_count, int T, int H, int B,
                           double uniform_prob, uint64_t seed, uint64_t offset, int64_t* ts, double* pt, double* p_out,
                           void* stream) {
    GD_CHECK_SHAPE(B > 0 && T > 0 && H > 0, "sample_timesteps: bad shape");
    GD_CHECK_ARG(T <= 4096, "sample_timesteps: T > 4096 unsupported");
    hipLaunchKernelGGL(sample_timesteps_kernel, dim3(1), dim3(256), (size_t)T * 16, (hipStream_t)stream, Lt_history,
                       Lt_count, T, H, B, uniform_prob, seed, offset, ts, pt, p_out, t_gd_step_state);
    return gd_launch_status("sample_timesteps");
}

static int adamw_launch(const int64_t* table, const int64_t* shadow_table, int n_tensors, int total_blocks, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                        void* stream) {
    GD_CHECK_ARG(n_tensors > 0 && total_blocks > 0 && step >= 1, "adamw: bad arguments");
    const AdamHyper h = gd_adam_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
    {
        // algorithmic bytes: read p, g, m, v; write p, m, v (+ the 2-byte shadow of p in bf16 mode)
        GdProfScope prof(6, (shadow_table ? 30.0 : 28.0) * ADAM_BLOCK_ELEMS * (double)total_blocks, (hipStream_t)stream);
        // nontemporal accesses for the streamed optimiser state: A/B on one box 0.344 -> 0.318 ms at the Yelp shape,
        // 1.032 -> 0.911 ms at the Amazon-Book shape (GDMCF_ADAMW_NT=0 switches back for comparison runs)
        static const bool nt = !(getenv("GDMCF_ADAMW_NT") && atoi(getenv("GDMCF_ADAMW_NT")) == 0);
        if (nt)
            hipLaunchKernelGGL(adamw_kernel<true>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table, n_tensors, h,
                               shadow_table, t_gd_step_state);
        else
            hipLaunchKernelGGL(adamw_kernel<false>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table, n_tensors,
                               h, shadow_table, t_gd_step_state);
    }
    return gd_launch_status("adamw");
}

int gdmcf_adamw_f32(const int64_t* table, int n_tensors, int total_blocks, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, float grad_scale, void* stream) {
    return adamw_launch(table, nullptr, n_tensors, total_blocks, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                        stream);
}

int gdmcf_adamw_bf16s_f32(const int64_t* table, const int64_t* shadow_table, int n_tensors, int total_blocks, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                          void* stream) {
    GD_CHECK_ARG(shadow_table != nullptr, "adamw_bf16s: shadow table missing");
    return adamw_launch(table, shadow_table, n_tensors, total_blocks, lr, beta1, beta2, eps, weight_decay, step,
                        grad_scale, stream);
}

int gdmcf_densify_rows_f32(const int64_t* indptr, const int32_t* indices, const float* values, const int64_t* row_ids,
                           int B, int I, float* out, int64_t ldo, void* stream) {
    GD_CHECK_SHAPE(B > 0 && I > 0 && ldo >= I, "densify_rows: bad shape");
    hipLaunchKernelGGL(densify_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, indptr, indices, values, row_ids,
                       I, out, ldo);
    return gd_launch_status("densify_rows");
}

int gdmcf_scale_f32(const float* acc, int64_t n, float scale, float* out, void* stream) {
    GD_CHECK_SHAPE(n > 0, "scale: empty");
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, acc, n,
                       scale, out);
    return gd_launch_status("scale");
}

}  // extern "C"

// The C ABI carries the hyper-parameters as float; torch.optim.AdamW forms its scalars from the Python doubles the user wrote
// (1 - 0.999 = 0.001, whereas 1 - (double)0.999f = 0.00099998713: 1.3e-5 off in every exp_avg_sq increment).  The double the
// user meant is the shortest decimal that rounds to the float we were given (7 significant digits identify a float).
static double gd_decimal(float x) {
    char buf[32];
    snprintf(buf, sizeof buf, "%.7g", (double)x);
    const double d = strtod(buf, nullptr);
    return (float)d == x ? d : (double)x;
}

GdAdamHyper gd_adam_hyper(float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale) {
    // scalars formed in double exactly as torch/optim/adamw.py does, then narrowed to f32
    GdAdamHyper h;
    const double lr_d = gd_decimal(lr), b1 = gd_decimal(beta1), b2 = gd_decimal(beta2), wd = gd_decimal(weight_decay);
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    h.decay = (float)(1.0 - lr_d * wd);
    h.one_m_b1 = (float)(1.0 - b1);
    h.beta2 = beta2;
    h.one_m_b2 = (float)(1.0 - b2);
    h.bc2_sqrt = (float)sqrt(bc2);
    h.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    h.eps = eps;
    h.neg_step = (float)(-(lr_d / bc1));
    h.grad_scale = grad_scale;
    return h;
}

// ---- graph step state ---------------------------------------------------------------------------------------------------
thread_local const GdStepState* t_gd_step_state = nullptr;

__global__ void graph_state_tick_kernel(GdStepState* st) {
    st->prep_offset += 1;
    st->ts_offset += 1;
    st->adam_step += 1;
    int64_t k = st->adam_step - st->table_first;
    k = k < 0 ? 0 : (k >= st->table_len ? st->table_len - 1 : k);  // the host refills the table before it runs out
    st->hyper = st->hyper_table[k];
}

extern "C" {

int gdmcf_graph_state_bytes(void) { return (int)sizeof(GdStepState); }
int gdmcf_adam_hyper_bytes(void) { return (int)sizeof(GdAdamHyper); }

int gdmcf_graph_state_bind(const void* state_dev) {
    t_gd_step_state = static_cast<const GdStepState*>(state_dev);
    return GDMCF_OK;
}

int gdmcf_graph_state_init(void* state_host, uint64_t prep_offset, uint64_t ts_offset, int64_t adam_step, int64_t table_first,
                           int64_t table_len, const void* hyper_table_dev) {
    GD_CHECK_ARG(state_host && hyper_table_dev && table_len > 0, "graph_state_init: null pointer / empty table");
    GdStepState* st = static_cast<GdStepState*>(state_host);
    st->prep_offset = prep_offset; st->ts_offset = ts_offset; st->adam_step = adam_step; st->table_first = table_first;
    st->table_len = table_len; st->hyper_table = static_cast<const GdAdamHyper*>(hyper_table_dev);
    st->hyper = GdAdamHyper{};
    return GDMCF_OK;
}

int gdmcf_adam_hyper_fill(void* out_host, int n, float lr, float beta1, float beta2, float eps, float weight_decay,
                          int64_t first_step, float grad_scale) {
    GD_CHECK_ARG(out_host && n > 0 && first_step >= 1, "adam_hyper_fill: bad arguments");
    GdAdamHyper* o = static_cast<GdAdamHyper*>(out_host);
    for (int k = 0; k < n; ++k) o[k] = gd_adam_hyper(lr, beta1, beta2, eps, weight_decay, (int)(first_step + k), grad_scale);
    return GDMCF_OK;
}

int gdmcf_graph_state_tick(void* state_dev, void* stream) {
    GD_CHECK_ARG(state_dev, "graph_state_tick: null state");
    hipLaunchKernelGGL(graph_state_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, static_cast<GdStepState*>(state_dev));
    return gd_launch_status("graph_state_tick");
}

}  // extern "C"

// ---- internal helpers used by linear.hip ---------------------------------------------------------
int gd_splitk_reduce(const float* slabs, int64_t slab_stride, int splits, int64_t ld_slab, int M, int N, int mode,
                     const float* bias, const float* rowscale, const float* aact, int64_t ldact, int act, float* out,
                     int64_t ldo, hipStream_t s) {
    const int64_t n = (int64_t)M * N;
    GdShadow sh;
    const bool has16 = gd_shadow_lookup(out, &sh) && sh.rows == M && sh.cols == N;
    (void)n;
    const int vec = (ld_slab % 4 == 0) && (slab_stride % 4 == 0) && gd_aligned16(slabs);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(gd_cdiv(N, 1024), M), dim3(256), 0, s, slabs, slab_stride,
                       splits, ld_slab, M, N, mode, bias, rowscale, aact, ldact, act, out, ldo,
                       has16 ? (unsigned short*)sh.p16 : nullptr, has16 ? sh.ld16 : 0, vec);
    return gd_launch_status("splitk_reduce");
}

int gd_colsum(const float* dZ, int64_t ld, const float* rs, int M, int N, float* db, hipStream_t s) {
    hipLaunchKernelGGL(colsum_kernel, dim3(gd_cdiv(N, 64)), dim3(256), 0, s, dZ, ld, rs, M, N, db);
    return gd_launch_status("colsum");
}

int gd_rowpart_reduce(const float* rowpart, int ld, int M, int nt, float* rowsum, hipStream_t s) {
    hipLaunchKernelGGL(rowpart_reduce_kernel, dim3(gd_cdiv(M, 4)), dim3(256), 0, s, rowpart, ld, M, nt, rowsum);
    return gd_launch_status("rowpart_reduce");
}

// ---- f32 -> bf16 shadow copy (gdmcf_bf16_shadow_sync; weights whose values changed outside the library) -------
namespace {
typedef __bf16 gd_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned gd_pack_bf16(float lo, float hi) {
    gd_bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}
typedef unsigned int gd_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, int64_t ld, unsigned short* __restrict__ dst,
                                                        int64_t ld16, int64_t rows, int64_t cols) {
    const int64_t groups = (cols + 7) / 8;  // 8 columns = one 16-byte store
    const int64_t total = rows * groups;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = u / groups, c = (u - r * groups) * 8;
        const float* p = src + r * ld + c;
        float e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = (c + i < cols) ? p[i] : 0.f;
        gd_u32x4 w;
        w.x = gd_pack_bf16(e[0], e[1]);
        w.y = gd_pack_bf16(e[2], e[3]);
        w.z = gd_pack_bf16(e[4], e[5]);
        w.w = gd_pack_bf16(e[6], e[7]);
        *reinterpret_cast<gd_u32x4*>(dst + r * ld16 + c) = w;
    }
}
}  // namespace

int gd_cast_bf16(const float* src, int64_t ld, void* dst, int64_t ld16, int64_t rows, int64_t cols, hipStream_t s) {
    const int64_t total = rows * ((cols + 7) / 8);
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, src, ld,
                       (unsigned short*)dst, ld16, rows, cols);
    return gd_launch_status("cast_bf16");
}
