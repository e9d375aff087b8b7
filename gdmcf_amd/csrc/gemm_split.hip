// float32 GEMM on the bf16 matrix pipe: every operand is split ON CHIP into three bfloat16 terms and the product is
// assembled from six bf16 MFMAs with f32 accumulation (GDMCF_GEMM_F32X3, DNN(gemm_dtype="f32x3")).
//
//   a = a0 + a1 + a2   with a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)   (round to nearest even; the
//   subtractions are exact in f32, three 8-bit significands cover the 24 bits of a float: |a - a0 - a1 - a2| <= 2^-26 |a|)
//   a*b ~= a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)
// The dropped terms (a1 b2, a2 b1, a2 b2) are below 2^-25 |a b|, i.e. under half an ulp of the product; every partial
// product of two bf16 values is exact in f32 and v_mfma_f32_16x16x32_bf16 accumulates in f32.  The result therefore
// carries f32-level error (tests/test_gpu_split.py compares it with float64 beside the native f32 MFMA path), at 6 x 16
// cycles per 16x16x32 block instead of 8 x 32 for v_mfma_f32_16x16x4_f32: 2.67x the matrix-pipe rate.  Non-finite inputs
// give NaN (inf - inf in the split).
//
// Same products, operand descriptions, tile classes, split-K scheme and epilogues as gemm_f32.hip.  The operands stay
// float32 in HBM; the split happens between the global loads and the LDS image, so no producer changes and nothing extra
// is stored.  Structure: k-tiles of 32, ONE LDS stage of three bf16 planes per operand (49 KB at 128x128: up to three
// workgroups per CU, whose convert / MFMA phases interleave on the SIMDs), loads of tile t+1 in flight in registers while
// tile t is multiplied.
//
// LDS image of a plane: rows of 32 bf16 = 64 bytes, four 16-byte slots, slot s of row r stored at s ^ (((r >> 3) & 1) << 1).
// ds_read_b128 serves 16-lane groups that hold 8 rows at k-slot q and 8 rows at q+1 (MI355X_MICROARCH.md, LDS): with this
// swizzle the 16 addresses fall on 16 distinct 16-byte columns of the 256-byte bank row.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BK = 32;
constexpr int PLANES = 3;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;  // v_cvt_pk_bf16_f32: round to nearest even
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// (x, y) -> three packed bf16 pairs
__device__ __forceinline__ void split2(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pack_bf16(x, y);
    const float rx = x - bf16_lo(p0), ry = y - bf16_hi(p0);
    p1 = pack_bf16(rx, ry);
    p2 = pack_bf16(rx - bf16_lo(p1), ry - bf16_hi(p1));
}

// byte offset of 16-byte slot `slot` (0..3) of row r inside a plane
__device__ __forceinline__ int img_off(int r, int slot) { return r * 64 + ((slot ^ (((r >> 3) & 1) << 1)) << 4); }

// ---- staging: global f32 -> registers -> three bf16 planes in LDS -----------------------------------------------------
// Units of an operand are dealt to threads [T0, T0 + UNITS) when it has fewer units than the workgroup has threads, so that
// the two operands of a product keep different waves busy (wave-uniform tests); otherwise thread t takes units t, t+NT, ...
template <int LAY, int R, int NT, int T0>
struct Stage;

// K-contiguous source [rows][K]: unit = (row, 16-byte segment of 4 floats), 8 per row
template <int R, int NT, int T0>
struct Stage<GD_LAY_KC, R, NT, T0> {
    static constexpr int UNITS = R * 8;
    static constexpr int NL = (UNITS + NT - 1) / NT;
    static constexpr int OFF = (UNITS < NT) ? T0 : 0;
    f32x4 reg[NL];

    __device__ static __forceinline__ bool fast(int row0, int rows_total, int k0, int kend) { return k0 + BK <= kend; }
    __device__ __forceinline__ void load_fast(const float* __restrict__ src, int64_t ld, int row0, int rows_total, int k0,
                                              int kend, int tid) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = max(tid - OFF, 0) + i * NT;
            const int row = min(row0 + min(u >> 3, R - 1), rows_total - 1);  // clamped: results of such rows are never stored
            reg[i] = *reinterpret_cast<const f32x4_u*>(src + (int64_t)row * ld + k0 + ((u & 7) << 2));
        }
    }
    // k-tail tile: nothing is read beyond kend and the tail contributes zeros
    __device__ __forceinline__ void load_safe(const float* __restrict__ src, int64_t ld, int row0, int rows_total, int k0,
                                              int kend, int tid) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = max(tid - OFF, 0) + i * NT;
            const int row = min(row0 + min(u >> 3, R - 1), rows_total - 1);
            const int k = k0 + ((u & 7) << 2);
            const float* p = src + (int64_t)row * ld + k;
            f32x4 v;
            v.x = (k + 0 < kend) ? p[0] : 0.f;
            v.y = (k + 1 < kend) ? p[1] : 0.f;
            v.z = (k + 2 < kend) ? p[2] : 0.f;
            v.w = (k + 3 < kend) ? p[3] : 0.f;
            reg[i] = v;
        }
    }
    __device__ __forceinline__ void store(char* img, int plane_bytes, int k0, int kend, int tid) const {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid - OFF + i * NT;
            if ((OFF != 0 || UNITS % NT != 0) && (u < 0 || u >= UNITS)) continue;
            const int row = u >> 3, seg = u & 7;
            unsigned a0, a1, a2, b0, b1, b2;
            split2(reg[i].x, reg[i].y, a0, a1, a2);
            split2(reg[i].z, reg[i].w, b0, b1, b2);
            char* p = img + img_off(row, seg >> 1) + ((seg & 1) << 3);
            *reinterpret_cast<u32x2*>(p) = u32x2{a0, b0};
            *reinterpret_cast<u32x2*>(p + plane_bytes) = u32x2{a1, b1};
            *reinterpret_cast<u32x2*>(p + 2 * plane_bytes) = u32x2{a2, b2};
        }
    }
};

// row-contiguous source [K][rows]: unit = (4-row group, 8-deep k group): an 8(k) x 4(rows) patch per lane, transposed
// in registers; four consecutive lanes hold the four k-groups of the same rows = one full 64-byte image row per plane
template <int R, int NT, int T0>
struct Stage<GD_LAY_MC, R, NT, T0> {
    static constexpr int UNITS = (R / 4) * 4;
    static constexpr int NL = (UNITS + NT - 1) / NT;
    static constexpr bool PART = UNITS < NT;  // only threads [OFF, OFF + UNITS) take part (NL == 1)
    static constexpr int OFF = PART ? T0 : 0;
    f32x4 reg[NL][8];

    __device__ static __forceinline__ bool fast(int row0, int rows_total, int k0, int kend) { return row0 + R <= rows_total; }
    // k rows beyond kend are read from the last valid row here and zeroed at store time
    __device__ __forceinline__ void load_fast(const float* __restrict__ src, int64_t ld, int row0, int rows_total, int k0,
                                              int kend, int tid) {
        if (PART && (tid < OFF || tid >= OFF + UNITS)) return;  // wave-uniform when OFF and UNITS are multiples of 64
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid - OFF + i * NT;
            const int kg = u & 3, rg = min(u >> 2, R / 4 - 1);
            const float* p = src + row0 + (rg << 2);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                reg[i][kk] = *reinterpret_cast<const f32x4_u*>(p + (int64_t)min(k0 + (kg << 3) + kk, kend - 1) * ld);
        }
    }
    // tile crossing the last row of the matrix: a 16-byte load could run past the row, go element-wise
    __device__ __forceinline__ void load_safe(const float* __restrict__ src, int64_t ld, int row0, int rows_total, int k0,
                                              int kend, int tid) {
        if (PART && (tid < OFF || tid >= OFF + UNITS)) return;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid - OFF + i * NT;
            const int kg = u & 3, rg = min(u >> 2, R / 4 - 1);
            const int row = row0 + (rg << 2);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float* p = src + (int64_t)min(k0 + (kg << 3) + kk, kend - 1) * ld + row;
                f32x4 v;
                v.x = (row + 0 < rows_total) ? p[0] : 0.f;
                v.y = (row + 1 < rows_total) ? p[1] : 0.f;
                v.z = (row + 2 < rows_total) ? p[2] : 0.f;
                v.w = (row + 3 < rows_total) ? p[3] : 0.f;
                reg[i][kk] = v;
            }
        }
    }
    __device__ __forceinline__ void store(char* img, int plane_bytes, int k0, int kend, int tid) const {
        if (PART && (tid < OFF || tid >= OFF + UNITS)) return;
        const bool ktail = k0 + BK > kend;  // workgroup-uniform
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid - OFF + i * NT;
            if (!PART && UNITS % NT != 0 && u >= UNITS) continue;
            const int kg = u & 3, rg = u >> 2;
            const int nvalid = ktail ? kend - (k0 + (kg << 3)) : 8;  // k values of this group inside [k0, kend)
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                float e[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) e[kk] = reg[i][kk][mm];
                if (ktail) {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) e[kk] = (kk < nvalid) ? e[kk] : 0.f;
                }
                unsigned w0[4], w1[4], w2[4];
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) split2(e[2 * pr], e[2 * pr + 1], w0[pr], w1[pr], w2[pr]);
                char* p = img + img_off((rg << 2) + mm, kg);
                *reinterpret_cast<u32x4*>(p) = u32x4{w0[0], w0[1], w0[2], w0[3]};
                *reinterpret_cast<u32x4*>(p + plane_bytes) = u32x4{w1[0], w1[1], w1[2], w1[3]};
                *reinterpret_cast<u32x4*>(p + 2 * plane_bytes) = u32x4{w2[0], w2[1], w2[2], w2[3]};
            }
        }
    }
};

template <class S>
__device__ __forceinline__ void stage_load(S& st, const float* __restrict__ src, int64_t ld, int row0, int rows_total, int k0,
                                           int kend, int tid) {
    if (S::fast(row0, rows_total, k0, kend))
        st.load_fast(src, ld, row0, rows_total, k0, kend, tid);
    else
        st.load_safe(src, ld, row0, rows_total, k0, kend, tid);
}

template <int LAYA, int LAYB, int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int MINWG>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MINWG) void gemm_split_kernel(const GdGemm g) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    static_assert(TM * 16 * WAVES_M == BM && TN * 16 * WAVES_N == BN, "tile must split into 16x16 blocks");
    constexpr int PLANE_BYTES = (BM + BN) * 64;  // A rows, then B rows
    constexpr int A_BYTES = BM * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WTM, wn0 = (wave % WAVES_N) * WTN;
    const int r = lane & 15, q = lane >> 4;

    // XCD-aware bijective remap (as gemm_f32.hip): consecutive logical tiles share an XCD's L2
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    const int tiles = g.tiles_m * g.tiles_n;
    const int split = logical / tiles;
    const int t = logical - split * tiles;
    const int tile_m = g.m_fastest ? (t % g.tiles_m) : (t / g.tiles_n);
    const int tile_n = g.m_fastest ? (t / g.tiles_m) : (t % g.tiles_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nt = (kend - kbeg + BK - 1) / BK;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operand B's units go to the upper threads when it has fewer units than threads (A's to the lower ones)
    Stage<LAYA, BM, NT, 0> sa;
    Stage<LAYB, BN, NT, (Stage<LAYB, BN, NT, 0>::UNITS < NT ? NT - Stage<LAYB, BN, NT, 0>::UNITS : 0)> sb;

    if (nt > 0) {
        stage_load(sa, g.A, g.lda, m0, g.M, kbeg, kend, tid);
        stage_load(sb, g.B, g.ldb, n0, g.N, kbeg, kend, tid);
    }
    for (int it = 0; it < nt; ++it) {
        const int k0 = kbeg + it * BK;
        sa.store(lds, PLANE_BYTES, k0, kend, tid);
        sb.store(lds + A_BYTES, PLANE_BYTES, k0, kend, tid);
        __syncthreads();
        if (it + 1 < nt) {  // in flight while this tile is multiplied
            stage_load(sa, g.A, g.lda, m0, g.M, k0 + BK, kend, tid);
            stage_load(sb, g.B, g.ldb, n0, g.N, k0 + BK, kend, tid);
        }
        bf16x8 fa[TM][PLANES];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < PLANES; ++p)
                fa[i][p] = *reinterpret_cast<const bf16x8*>(lds + p * PLANE_BYTES + img_off(wm0 + 16 * i + r, q));
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bf16x8 fb[PLANES];
#pragma unroll
            for (int p = 0; p < PLANES; ++p)
                fb[p] = *reinterpret_cast<const bf16x8*>(lds + p * PLANE_BYTES + A_BYTES + img_off(wn0 + 16 * j + r, q));
            // smallest terms first
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][2], fb[0], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[2], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[1], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[0], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[1], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[0], acc[i][j], 0, 0, 0);
        }
        __syncthreads();  // every wave has read this tile before the next one overwrites it
    }
    if constexpr (EPI == GD_EPI_ADAMW)
        gemm_epilogue_rows<BM, BN, TM, TN, WAVES_M, WAVES_N, EPI, PLANES * PLANE_BYTES / 4, NT>(acc, g, m0, n0, wn0, r, q, wave,
                                                                                                tid, smem);
    else
        gemm_epilogue<BM, TM, TN, WAVES_N, EPI>(acc, g, m0, n0, wm0, wn0, r, q, split, tile_n, wave, tid, smem);
}

// ---- wave-specialised variant ---------------------------------------------------------------------------------------
// 512 threads: waves 0-3 issue nothing but LDS fragment reads and MFMAs (96 per k-tile at 128x128), waves 4-7 bring the
// operands in: global loads into a ring of NSTG register stages (asm loads + counted s_waitcnt: tiles it+1 .. it+NSTG-1 are
// in flight while tile it is multiplied), the three-term split, the LDS writes.  The split costs ~4.5 vector instructions per
// element; in loader waves they issue in the shadow of the MFMA waves' matrix instructions (an MFMA holds a SIMD's vector
// issue for 8 of its 16 cycles).  LDS is double buffered (2 x 49 KB at 128x128: one workgroup per CU), one s_barrier per
// k-tile for all eight waves.  The register ring takes whole interior k-tiles only; the K tail and workgroups whose
// row-contiguous operand crosses the matrix edge go through the synchronous Stage path above, same LDS image.
template <int LAY, int R, int NT, int T0>
struct RingStage : Stage<LAY, R, NT, T0> {
    using S = Stage<LAY, R, NT, T0>;
    static constexpr int LOADS = (LAY == GD_LAY_KC) ? S::NL : S::NL * 8;
    static __device__ __forceinline__ f32x4 gload(const float* p) {
        f32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    // interior tile only (k0 + BK <= kend; row-contiguous: row0 + R <= rows_total); always LOADS loads per thread
    __device__ __forceinline__ void load_ring(const float* __restrict__ src, int64_t ld, int row0, int rows_total, int k0,
                                              int tid) {
        if constexpr (LAY == GD_LAY_KC) {
#pragma unroll
            for (int i = 0; i < S::NL; ++i) {
                const int u = tid + i * NT;
                const int row = min(row0 + min(u >> 3, R - 1), rows_total - 1);
                this->reg[i] = gload(src + (int64_t)row * ld + k0 + ((u & 7) << 2));
            }
        } else {
#pragma unroll
            for (int i = 0; i < S::NL; ++i) {
                const int u = min(max(tid - S::OFF, 0) + i * NT, S::UNITS - 1);  // idle threads load a valid patch too
                const int kg = u & 3, rg = u >> 2;
                const float* p = src + (int64_t)(k0 + (kg << 3)) * ld + row0 + (rg << 2);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) this->reg[i][kk] = gload(p + (int64_t)kk * ld);
            }
        }
    }
    __device__ __forceinline__ void pin() {
        if constexpr (LAY == GD_LAY_KC) {
#pragma unroll
            for (int i = 0; i < S::NL; ++i) asm volatile("" : "+v"(this->reg[i]));
        } else {
#pragma unroll
            for (int i = 0; i < S::NL; ++i)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(this->reg[i][kk]));
        }
    }
};

// Both operands row-contiguous (the weight gradients): one 8(k) x 4(rows) patch per loader thread and ring slot -- threads
// [0, BM) take A's patches, [BM, BM + BN) B's (BM/4 row groups x 4 k-groups each) -- so a slot costs 32 registers.
template <int BM, int BN, int NT>
struct McPair {
    static_assert(BM + BN <= NT, "one patch per loader thread");
    static constexpr int LOADS = 8;
    f32x4 reg[8];
    __device__ __forceinline__ void load_ring(const GdGemm& g, int m0, int n0, int k0, int tid) {
        const bool isA = tid < BM;
        const int u = isA ? tid : min(tid - BM, BN - 1);
        const int kg = u & 3, rg = u >> 2;
        const int64_t ld = isA ? g.lda : g.ldb;
        const float* p = (isA ? g.A + m0 : g.B + n0) + (int64_t)(k0 + (kg << 3)) * ld + (rg << 2);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(reg[kk]) : "v"(p) : "memory");
            p += ld;
        }
    }
    __device__ __forceinline__ void pin() {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(reg[kk]));
    }
    __device__ __forceinline__ void store(char* img, int plane_bytes, int tid) const {
        if (tid >= BM + BN) return;
        const bool isA = tid < BM;
        const int u = isA ? tid : tid - BM;
        const int kg = u & 3, rg = u >> 2;
        char* base = img + (isA ? 0 : BM * 64);
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            unsigned w0[4], w1[4], w2[4];
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) split2(reg[2 * pr][mm], reg[2 * pr + 1][mm], w0[pr], w1[pr], w2[pr]);
            char* p = base + img_off((rg << 2) + mm, kg);
            *reinterpret_cast<u32x4*>(p) = u32x4{w0[0], w0[1], w0[2], w0[3]};
            *reinterpret_cast<u32x4*>(p + plane_bytes) = u32x4{w1[0], w1[1], w1[2], w1[3]};
            *reinterpret_cast<u32x4*>(p + 2 * plane_bytes) = u32x4{w2[0], w2[1], w2[2], w2[3]};
        }
    }
};

// one ring slot: both operands' registers of one k-tile
template <int LAYA, int LAYB, int BM, int BN, int NT>
struct RingSlot {
    static constexpr bool PAIR = (LAYA == GD_LAY_MC && LAYB == GD_LAY_MC && BM + BN <= NT);
    using SA = RingStage<LAYA, BM, NT, 0>;
    using SB = RingStage<LAYB, BN, NT, (Stage<LAYB, BN, NT, 0>::UNITS < NT ? NT - Stage<LAYB, BN, NT, 0>::UNITS : 0)>;
    struct Two {
        SA a;
        SB b;
    };
    typename std::conditional<PAIR, McPair<BM, BN, NT>, Two>::type s;
    static constexpr int LOADS = PAIR ? 8 : SA::LOADS + SB::LOADS;
    __device__ __forceinline__ void load_ring(const GdGemm& g, int m0, int n0, int k0, int tid) {
        if constexpr (PAIR) {
            s.load_ring(g, m0, n0, k0, tid);
        } else {
            s.a.load_ring(g.A, g.lda, m0, g.M, k0, tid);
            s.b.load_ring(g.B, g.ldb, n0, g.N, k0, tid);
        }
    }
    __device__ __forceinline__ void pin() {
        if constexpr (PAIR) {
            s.pin();
        } else {
            s.a.pin();
            s.b.pin();
        }
    }
    __device__ __forceinline__ void store(char* img, int plane_bytes, int k0, int kend, int tid) const {
        if constexpr (PAIR) {
            s.store(img, plane_bytes, tid);
        } else {
            s.a.store(img, plane_bytes, k0, kend, tid);
            s.b.store(img + BM * 64, plane_bytes, k0, kend, tid);
        }
    }
};

template <int LAYA, int LAYB, int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int NSTG>
__global__ __launch_bounds__(512, 2) void gemm_split_spec_kernel(const GdGemm g) {
    static_assert(WAVES_M * WAVES_N == 4, "four MFMA waves");
    constexpr int NT = 256;  // threads of either role
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    static_assert(TM * 16 * WAVES_M == BM && TN * 16 * WAVES_N == BN, "tile must split into 16x16 blocks");
    constexpr int PLANE_BYTES = (BM + BN) * 64;
    constexpr int A_BYTES = BM * 64;
    constexpr int STAGE_BYTES = PLANES * PLANE_BYTES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const bool loader = threadIdx.x >= NT;
    const int tid = threadIdx.x & (NT - 1), lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WTM, wn0 = (wave % WAVES_N) * WTN;
    const int r = lane & 15, q = lane >> 4;

    const int nwg = gridDim.x, id = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    const int tiles = g.tiles_m * g.tiles_n;
    const int split = logical / tiles;
    const int t = logical - split * tiles;
    const int tile_m = g.m_fastest ? (t % g.tiles_m) : (t / g.tiles_n);
    const int tile_n = g.m_fastest ? (t / g.tiles_m) : (t % g.tiles_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nt = (kend - kbeg + BK - 1) / BK;

    if (g.dbg & (loader ? 32 : 16)) __builtin_amdgcn_s_setprio(1);  // ablation: issue priority for one of the two roles
    if (loader) {
        using Slot = RingSlot<LAYA, LAYB, BM, BN, NT>;
        Slot ring[NSTG];
        constexpr int LPT = Slot::LOADS;
        static_assert(NSTG >= 2 && (NSTG - 2) * LPT <= 63, "vmcnt is a 6-bit counter");
#define GD_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
        auto wait_tiles = [&](int after) {  // every load but those of the `after` youngest tiles has landed
            switch (after) {
                case 0: GD_WAIT_VM(0); break;
                case 1: GD_WAIT_VM(LPT); break;
                case 2: GD_WAIT_VM(2 * LPT < 64 ? 2 * LPT : 0); break;
                default: GD_WAIT_VM(3 * LPT < 64 ? 3 * LPT : 0); break;
            }
        };
        const bool rows_ok = (LAYA == GD_LAY_KC || m0 + BM <= g.M) && (LAYB == GD_LAY_KC || n0 + BN <= g.N);
        const int ring_end = rows_ok ? (kend - kbeg) / BK : 0;  // tiles [0, ring_end) come through the register ring
        // a tile outside the ring (K tail; every tile of a workgroup on the matrix edge): synchronously, compiler-counted loads
        auto sync_tile = [&](int tile, char* img) {
            Stage<LAYA, BM, NT, 0> a;
            typename Slot::SB::S b;
            const int k0 = kbeg + tile * BK;
            stage_load(a, g.A, g.lda, m0, g.M, k0, kend, tid);
            stage_load(b, g.B, g.ldb, n0, g.N, k0, kend, tid);
            a.store(img, PLANE_BYTES, k0, kend, tid);
            b.store(img + A_BYTES, PLANE_BYTES, k0, kend, tid);
        };
#pragma unroll
        for (int u = 0; u < NSTG - 1; ++u)
            if (u < ring_end) ring[u].load_ring(g, m0, n0, kbeg + u * BK, tid);
        if (nt > 0) {
            if (ring_end > 0) {
                wait_tiles(min(NSTG - 2, ring_end - 1));
                ring[0].pin();
                ring[0].store(lds, PLANE_BYTES, kbeg, kend, tid);
            } else {
                sync_tile(0, lds);
            }
        }
        __syncthreads();
        for (int base = 0; base < nt; base += NSTG) {
#pragma unroll
            for (int u = 0; u < NSTG; ++u) {
                const int it = base + u;  // the MFMA waves multiply tile `it`; this step makes tile it+1 visible in LDS
                if (it >= nt) break;
                const int s_new = (u + NSTG - 1) % NSTG, s_nxt = (u + 1) % NSTG;
                if (it + NSTG - 1 < ring_end && !(g.dbg & 4)) ring[s_new].load_ring(g, m0, n0, kbeg + (it + NSTG - 1) * BK, tid);
                if (it + 1 < nt) {
                    char* Ln = lds + ((it + 1) & 1) * STAGE_BYTES;
                    if (it + 1 < ring_end) {
                        wait_tiles(min(it + NSTG - 1, ring_end - 1) - (it + 1));
                        ring[s_nxt].pin();
                        if (!(g.dbg & 8)) ring[s_nxt].store(Ln, PLANE_BYTES, kbeg + (it + 1) * BK, kend, tid);
                    } else {
                        GD_WAIT_VM(0);
                        sync_tile(it + 1, Ln);
                    }
                }
                __syncthreads();
            }
        }
        GD_WAIT_VM(0);
#undef GD_WAIT_VM
        return;  // loader waves take no part in the epilogue (a finished wave leaves the barrier count)
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        const char* cur = lds + (it & 1) * STAGE_BYTES;
#define GD_SIX(AC, FA, FB) /* smallest terms first */                                              \
    AC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[2], FB[0], AC, 0, 0, 0);                       \
    AC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0], FB[2], AC, 0, 0, 0);                       \
    AC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[1], FB[1], AC, 0, 0, 0);                       \
    AC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[1], FB[0], AC, 0, 0, 0);                       \
    AC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0], FB[1], AC, 0, 0, 0);                       \
    AC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0], FB[0], AC, 0, 0, 0);
        // B's fragments stay resident for the tile; A's come block row by block row through a three-deep register ring:
        // the reads of block row i+3 are issued behind the MFMAs of block row i, so only the tile's first reads wait.
        // sched_barrier pins that order (left alone, hipcc groups the work by plane and waits for every pair of reads).
        bf16x8 fb[TN][PLANES], fa[3][PLANES];
        auto read_a = [&](int i, bf16x8 (&dst)[PLANES]) {
#pragma unroll
            for (int p = 0; p < PLANES; ++p)
                dst[p] = *reinterpret_cast<const bf16x8*>(cur + p * PLANE_BYTES + img_off(wm0 + 16 * i + r, q));
        };
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < PLANES; ++p)
                fb[j][p] = *reinterpret_cast<const bf16x8*>(cur + p * PLANE_BYTES + A_BYTES + img_off(wn0 + 16 * j + r, q));
#pragma unroll
        for (int i = 0; i < 3 && i < TM; ++i) read_a(i, fa[i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (!(g.dbg & 2)) {
#pragma unroll
                for (int j = 0; j < TN; ++j) { GD_SIX(acc[i][j], fa[i % 3], fb[j]) }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i + 3 < TM) read_a(i + 3, fa[i % 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef GD_SIX
        __syncthreads();
    }
    if constexpr (EPI == GD_EPI_ADAMW)
        gemm_epilogue_rows<BM, BN, TM, TN, WAVES_M, WAVES_N, EPI, 2 * STAGE_BYTES / 4, NT>(acc, g, m0, n0, wn0, r, q, wave, tid,
                                                                                          smem);
    else
        gemm_epilogue<BM, TM, TN, WAVES_N, EPI>(acc, g, m0, n0, wm0, wn0, r, q, split, tile_n, wave, tid, smem);
}

template <int LAYA, int LAYB, int BM, int BN, int WM, int WN, int EPI>
int launch_spec(GdGemm& g, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * PLANES * (BM + BN) * 64;
    // ring depth: three slots while they fit the loader waves' registers beside the split's temporaries, else two
    using Slot = RingSlot<LAYA, LAYB, BM, BN, 256>;
    constexpr int NSTG = (Slot::LOADS * 4 * 3 <= 150) ? 3 : 2;
    auto kern = gemm_split_spec_kernel<LAYA, LAYB, BM, BN, WM, WN, EPI, NSTG>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) {
            gdmcf_set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e));
            return GDMCF_E_HIP;
        }
        attr_set = true;
    }
    g.tiles_m = gd_cdiv(g.M, BM);
    g.tiles_n = gd_cdiv(g.N, BN);
    if (g.splits < 1) g.splits = 1;
    if (g.kchunk <= 0) g.kchunk = gd_cdiv(gd_cdiv(g.K, g.splits), BK) * BK;
    const long grid = (long)g.tiles_m * g.tiles_n * g.splits;
    if (grid <= 0 || grid > 0x7fffffffL) {
        gdmcf_set_error("gemm grid out of range: %ld", grid);
        return GDMCF_E_SHAPE;
    }
    {
        GdProfScope prof(g.prof_tag, 2.0 * g.M * g.N * g.K, s);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, g);
    }
    return gd_launch_status("gemm_split_spec");
}

template <int LAYA, int LAYB, int BM, int BN, int WM, int WN, int EPI, int MINWG>
int launch_one(GdGemm& g, hipStream_t s) {
    constexpr size_t lds = (size_t)PLANES * (BM + BN) * 64;
    auto kern = gemm_split_kernel<LAYA, LAYB, BM, BN, WM, WN, EPI, MINWG>;
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) {
            gdmcf_set_error("hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e));
            return GDMCF_E_HIP;
        }
        attr_set = true;
    }
    g.tiles_m = gd_cdiv(g.M, BM);
    g.tiles_n = gd_cdiv(g.N, BN);
    if (g.splits < 1) g.splits = 1;
    if (g.kchunk <= 0) g.kchunk = gd_cdiv(gd_cdiv(g.K, g.splits), BK) * BK;
    const long grid = (long)g.tiles_m * g.tiles_n * g.splits;
    if (grid <= 0 || grid > 0x7fffffffL) {
        gdmcf_set_error("gemm grid out of range: %ld", grid);
        return GDMCF_E_SHAPE;
    }
    {
        GdProfScope prof(g.prof_tag, 2.0 * g.M * g.N * g.K, s);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * WM * WN), lds, s, g);
    }
    return gd_launch_status("gemm_split");
}

template <int LAYA, int LAYB, int EPI>
int launch_class(int cls, GdGemm& g, hipStream_t s) {
    static const bool spec = !(getenv("GDMCF_SPLIT_SPEC") && atoi(getenv("GDMCF_SPLIT_SPEC")) == 0);  // A/B knob
    switch (cls) {
        case 0: return spec ? launch_spec<LAYA, LAYB, 80, 128, 1, 4, EPI>(g, s) : launch_one<LAYA, LAYB, 80, 128, 1, 4, EPI, 2>(g, s);
        case 1: return spec ? launch_spec<LAYA, LAYB, 128, 128, 2, 2, EPI>(g, s) : launch_one<LAYA, LAYB, 128, 128, 2, 2, EPI, 2>(g, s);
        case 2: return launch_one<LAYA, LAYB, 64, 64, 2, 2, EPI, 2>(g, s);
        case 4: return launch_spec<LAYA, LAYB, 208, 128, 1, 4, EPI>(g, s);
    }
    gdmcf_set_error("bad gemm shape class %d", cls);
    return GDMCF_E_ARG;
}

}  // namespace

int gd_gemm_split_launch(int layA, int layB, int epi, int cls, GdGemm& g, hipStream_t s) {
    static const int dbg = getenv("GDMCF_SPLIT_DBG") ? atoi(getenv("GDMCF_SPLIT_DBG")) : 0;
    g.dbg = dbg;
    if (g.kchunk > 0 && g.kchunk % BK != 0) {
        gdmcf_set_error("gemm_split: kchunk %d is not a multiple of %d", g.kchunk, BK);
        return GDMCF_E_ARG;
    }
    if (layA == GD_LAY_KC && layB == GD_LAY_KC) {
        switch (epi) {
            case GD_EPI_SLAB: return launch_class<GD_LAY_KC, GD_LAY_KC, GD_EPI_SLAB>(cls, g, s);
            case GD_EPI_BIAS_ACT: return launch_class<GD_LAY_KC, GD_LAY_KC, GD_EPI_BIAS_ACT>(cls, g, s);
            case GD_EPI_LOSS: return launch_class<GD_LAY_KC, GD_LAY_KC, GD_EPI_LOSS>(cls, g, s);
            case GD_EPI_POST: return launch_class<GD_LAY_KC, GD_LAY_KC, GD_EPI_POST>(cls, g, s);
        }
    } else if (layA == GD_LAY_KC && layB == GD_LAY_MC) {
        if (epi == GD_EPI_SLAB) return launch_class<GD_LAY_KC, GD_LAY_MC, GD_EPI_SLAB>(cls, g, s);
    } else if (layA == GD_LAY_MC && layB == GD_LAY_MC) {
        if (epi == GD_EPI_STORE) return launch_class<GD_LAY_MC, GD_LAY_MC, GD_EPI_STORE>(cls, g, s);
        if (epi == GD_EPI_ADAMW) return launch_class<GD_LAY_MC, GD_LAY_MC, GD_EPI_ADAMW>(cls, g, s);
    }
    gdmcf_set_error("unsupported split gemm variant (layA=%d layB=%d epi=%d)", layA, layB, epi);
    return GDMCF_E_UNSUPPORTED;
}
