// bf16-input GEMM core on v_mfma_f32_16x16x32_bf16 (BASELINE config "bf16 denoiser GEMM on MFMA").
//
// Same products, operand descriptions, tile classes, split-K scheme and epilogues as gemm_f32.hip; only the
// arithmetic of the inner product changes: both operands are bfloat16 (round-to-nearest-even of the float32
// values) multiplied on the bf16 matrix pipe with f32 accumulation.  The float32 tensors in HBM stay authoritative
// (master weights, activations, gradients, optimiser state).  Two operand sources, same results bit for bit:
//   * bf16 shadows (gdmcf_bf16_shadow_set, kept in sync by the producing kernels): streamed as they are, no
//     conversion, half the bytes; their zero padding replaces every K-tail / edge predicate (Stage16);
//   * the float32 tensors themselves, rounded with v_cvt_pk_bf16_f32 on the way to LDS (Stage) -- the fallback when
//     an operand has no shadow.
//
// LDS image (both operands, whatever their global layout): rows of 64 bf16 = 128 bytes, eight 16-byte slots,
// slot s of row r stored at s ^ ((r >> 1) & 7) -- byte-for-byte the access pattern of gemm_f32.hip's
// K-contiguous image, so its bank analysis (MI355X_MICROARCH.md, LDS) carries over: the ds_read_b128 of MFMA
// block t, k-chunk c by lane (r = lane & 15, q = lane >> 4) takes slot 4c+q of row 16t+r = that lane's eight
// consecutive k values, exactly one 16x16x32 operand.
//  * K-contiguous source: 16 lanes x 16 B cover one row's 64 floats; each lane converts 4 values and writes
//    8 bytes (ds_write_b64, 16 contiguous lanes = one full 128-byte row, conflict-free).
//  * row-contiguous source ([k][rows]): a lane loads a 8(k) x 4(rows) patch with eight 16-byte loads (lanes
//    along the rows: coalesced), transposes it in registers and writes four 16-byte slots; eight consecutive
//    lanes hold the eight k-groups of the same rows, i.e. write one full 128-byte row per ds_write_b128 group.
// With bf16 MFMA time per tile 16x smaller than f32, the kernel is bound by bringing the f32 sources in: at
// 80x128 tiles the L2 serves 14 TB/s of tile reads (rocprofv3 TCC_REQ) for 2.6 TB/s of HBM fetches.  Hence the
// plain structure (register-staged double buffering, one barrier per k-tile, occupancy to cover latency) and
// the extra tile class 3 = 208x256 on eight waves for the batch-sized products (M = 400 = 2 x 208 - 16): 55 %
// fewer L2 bytes per FLOP than 80x128.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BK = 64;

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

// byte offset of 16-byte slot `slot` of row r
__device__ __forceinline__ int img_off(int r, int slot) { return r * 128 + ((slot ^ ((r >> 1) & 7)) << 4); }

// ---- staging: global f32 -> registers -> bf16 LDS image ---------------------------------------------------
template <int LAY, int R, int NT>
struct Stage;

// Loads come in two flavours chosen per tile by a workgroup-uniform test, so that the common one is a run of
// back-to-back 16-byte loads without a branch, predicate or select between them (a select right after a load
// makes hipcc wait for that load before issuing the next):
//   fast()  true  -> load_fast: whole tile inside the matrix in the vectorised direction
//   fast()  false -> load_safe: element-wise predicated loads (edge tiles only)

// K-contiguous source [rows][K]
template <int R, int NT>
struct Stage<GD_LAY_KC, R, NT> {
    static constexpr int UNITS = R * 16;  // (row, 16-byte segment)
    static constexpr int NL = (UNITS + NT - 1) / NT;
    f32x4 reg[NL];

    __device__ static __forceinline__ bool fast(int row0, int rows_total, int k0, int kend) { return k0 + BK <= kend; }
    // rows are clamped into the matrix (results of out-of-range rows are never stored)
    __device__ __forceinline__ void load_fast(const float* __restrict__ src, int64_t ld, int row0, int rows_total,
                                              int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid + i * NT;
            const int row = min(row0 + min(u >> 4, R - 1), rows_total - 1);
            reg[i] = *reinterpret_cast<const f32x4_u*>(src + (int64_t)row * ld + k0 + ((u & 15) << 2));
        }
    }
    // k-tail tile: nothing is read beyond kend and the tail contributes zeros
    __device__ __forceinline__ void load_safe(const float* __restrict__ src, int64_t ld, int row0, int rows_total,
                                              int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid + i * NT;
            const int row = min(row0 + min(u >> 4, R - 1), rows_total - 1);
            const int k = k0 + ((u & 15) << 2);
            const float* p = src + (int64_t)row * ld + k;
            f32x4 v;
            v.x = (k + 0 < kend) ? p[0] : 0.f;
            v.y = (k + 1 < kend) ? p[1] : 0.f;
            v.z = (k + 2 < kend) ? p[2] : 0.f;
            v.w = (k + 3 < kend) ? p[3] : 0.f;
            reg[i] = v;
        }
    }
    __device__ __forceinline__ void store(char* img, int k0, int kend, int tid) const {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid + i * NT;
            if (UNITS % NT != 0 && u >= UNITS) continue;
            const int row = u >> 4, seg = u & 15;
            u32x2 w;
            w.x = pack_bf16(reg[i].x, reg[i].y);
            w.y = pack_bf16(reg[i].z, reg[i].w);
            *reinterpret_cast<u32x2*>(img + img_off(row, seg >> 1) + ((seg & 1) << 3)) = w;
        }
    }
};

// row-contiguous source [K][rows]
template <int R, int NT>
struct Stage<GD_LAY_MC, R, NT> {
    static constexpr int UNITS = (R / 4) * 8;  // (4-row group, 8-deep k group)
    static constexpr int NL = (UNITS + NT - 1) / NT;
    f32x4 reg[NL][8];

    __device__ static __forceinline__ bool fast(int row0, int rows_total, int k0, int kend) { return row0 + R <= rows_total; }
    // k rows beyond kend are read from the last valid row here and zeroed at store time
    __device__ __forceinline__ void load_fast(const float* __restrict__ src, int64_t ld, int row0, int rows_total,
                                              int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid + i * NT;
            const int kg = u & 7, rg = min(u >> 3, R / 4 - 1);
            const float* p = src + row0 + (rg << 2);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                reg[i][kk] = *reinterpret_cast<const f32x4_u*>(p + (int64_t)min(k0 + (kg << 3) + kk, kend - 1) * ld);
        }
    }
    // tile crossing the last row of the matrix: a 16-byte load could run past the row, go element-wise
    __device__ __forceinline__ void load_safe(const float* __restrict__ src, int64_t ld, int row0, int rows_total,
                                              int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid + i * NT;
            const int kg = u & 7, rg = min(u >> 3, R / 4 - 1);
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
    __device__ __forceinline__ void store(char* img, int k0, int kend, int tid) const {
        const bool ktail = k0 + BK > kend;  // workgroup-uniform
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid + i * NT;
            if (UNITS % NT != 0 && u >= UNITS) continue;
            const int kg = u & 7, rg = u >> 3;
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
                u32x4 w;
                w.x = pack_bf16(e[0], e[1]);
                w.y = pack_bf16(e[2], e[3]);
                w.z = pack_bf16(e[4], e[5]);
                w.w = pack_bf16(e[6], e[7]);
                *reinterpret_cast<u32x4*>(img + img_off((rg << 2) + mm, kg)) = w;
            }
        }
    }
};

template <int LAY, int R, int NT>
__device__ __forceinline__ void stage_load(Stage<LAY, R, NT>& st, const float* __restrict__ src, int64_t ld, int row0,
                                           int rows_total, int k0, int kend, int tid) {
    if (Stage<LAY, R, NT>::fast(row0, rows_total, k0, kend))
        st.load_fast(src, ld, row0, rows_total, k0, kend, tid);
    else
        st.load_safe(src, ld, row0, rows_total, k0, kend, tid);
}

// ---- staging from bf16 shadows (gdmcf_bf16_shadow_set): no conversion, half the bytes ------------------------
// Shadows are zero outside [rows, cols) up to ld16 (a multiple of 64 columns) and up to the next multiple of 64
// rows, so neither K tails nor edge tiles need predicates: every load is an unconditional 16-byte load.
template <int LAY, int R, int NT>
struct Stage16;

// Shadow tiles are loaded by inline asm so that hipcc neither counts nor waits for them: the shadow path keeps TWO tiles
// in flight in registers (its products are bound by the latency of the operand stream, ~2 us under load, not by the
// 16x-faster matrix pipe) and places the counted s_waitcnt vmcnt(N) itself; pin() makes the registers opaque behind it.
__device__ __forceinline__ u32x4 gload16_u(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int R, int NT>
struct Stage16<GD_LAY_KC, R, NT> {
    static constexpr int UNITS = R * 8;  // (row, 16-byte segment = 8 bf16)
    static constexpr int NL = (UNITS + NT - 1) / NT;
    u32x4 reg[NL];
    __device__ __forceinline__ void load(const unsigned short* __restrict__ src, int64_t ld, int row0, int rows_total,
                                         int k0, int tid) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid + i * NT;
            const int row = min(row0 + min(u >> 3, R - 1), rows_total - 1);
            reg[i] = *reinterpret_cast<const u32x4*>(src + (int64_t)row * ld + k0 + ((u & 7) << 3));
        }
    }
    // the same loads as inline asm (uncounted by hipcc): the two-stage ring of the K-contiguous x K-contiguous products
    __device__ __forceinline__ void load_asm(const unsigned short* __restrict__ src, int64_t ld, int row0, int rows_total,
                                             int k0, int tid) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid + i * NT;
            const int row = min(row0 + min(u >> 3, R - 1), rows_total - 1);
            reg[i] = gload16_u(src + (int64_t)row * ld + k0 + ((u & 7) << 3));
        }
    }
    static constexpr int LOADS = NL;
    __device__ __forceinline__ void pin() {
#pragma unroll
        for (int i = 0; i < NL; ++i) asm volatile("" : "+v"(reg[i]));
    }
    __device__ __forceinline__ void store(char* img, int tid) const {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid + i * NT;
            if (UNITS % NT != 0 && u >= UNITS) continue;
            *reinterpret_cast<u32x4*>(img + img_off(u >> 3, u & 7)) = reg[i];
        }
    }
};

template <int R, int NT>
struct Stage16<GD_LAY_MC, R, NT> {
    static constexpr int UNITS = R;  // (8-row group, 8-deep k group): R/8 * 8
    static constexpr int NL = (UNITS + NT - 1) / NT;
    u32x4 reg[NL][8];
    __device__ __forceinline__ void load(const unsigned short* __restrict__ src, int64_t ld, int row0, int rows_total,
                                         int k0, int tid) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid + i * NT;
            const int kg = u & 7, rg = min(u >> 3, R / 8 - 1);
            const int col = min(row0 + (rg << 3), (int)ld - 8);  // groups beyond the matrix read padding of this row
            const unsigned short* p = src + (int64_t)(k0 + (kg << 3)) * ld + col;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) reg[i][kk] = *reinterpret_cast<const u32x4*>(p + (int64_t)kk * ld);
        }
    }
    // 8x8 transpose of 16-bit values: image row (rg*8 + mm) slot kg holds k = kg*8 .. kg*8+7 of source row mm
    __device__ __forceinline__ void store(char* img, int tid) const {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int u = tid + i * NT;
            if (UNITS % NT != 0 && u >= UNITS) continue;
            const int kg = u & 7, rg = u >> 3;
#pragma unroll
            for (int mm = 0; mm < 8; ++mm) {
                const int d = mm >> 1;
                const unsigned sel = (mm & 1) ? 0x07060302u : 0x05040100u;
                u32x4 w;
                w.x = __builtin_amdgcn_perm(reg[i][1][d], reg[i][0][d], sel);
                w.y = __builtin_amdgcn_perm(reg[i][3][d], reg[i][2][d], sel);
                w.z = __builtin_amdgcn_perm(reg[i][5][d], reg[i][4][d], sel);
                w.w = __builtin_amdgcn_perm(reg[i][7][d], reg[i][6][d], sel);
                *reinterpret_cast<u32x4*>(img + img_off((rg << 3) + mm, kg)) = w;
            }
        }
    }
};

// Both shadows row-contiguous (the weight gradients): ONE (8-row group, 8-deep k group) unit per thread and ring slot --
// threads [0, BM) take A's units, the top BN threads B's -- instead of both operands' units piled on the low threads:
// 32 staging registers per slot (two slots fit), and every wave stages.
template <int BM, int BN, int NT>
struct Stage16Pair {
    static_assert(BM + BN <= NT, "one unit per thread");
    static constexpr int LOADS = 8;
    u32x4 reg[8];
    __device__ __forceinline__ void load_asm(const GdGemm& g, int m0, int n0, int k0, int tid) {
        const bool isA = tid < BM;
        const int u = isA ? tid : max(tid - (NT - BN), 0);
        const int kg = u & 7, rg = u >> 3;
        const int64_t ld = isA ? g.lda16 : g.ldb16;
        const int col = min((isA ? m0 : n0) + (rg << 3), (int)ld - 8);  // groups beyond the matrix read padding of this row
        const unsigned short* p = static_cast<const unsigned short*>(isA ? g.A16 : g.B16) + (int64_t)(k0 + (kg << 3)) * ld + col;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) reg[kk] = gload16_u(p + (int64_t)kk * ld);
    }
    __device__ __forceinline__ void pin() {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(reg[kk]));
    }
    __device__ __forceinline__ void store(char* img, int tid) const {
        const bool isA = tid < BM;
        if (!isA && tid < NT - BN) return;
        const int u = isA ? tid : tid - (NT - BN);
        const int kg = u & 7, rg = u >> 3;
        char* base = img + (isA ? 0 : BM * 128);
#pragma unroll
        for (int mm = 0; mm < 8; ++mm) {
            const int d = mm >> 1;
            const unsigned sel = (mm & 1) ? 0x07060302u : 0x05040100u;
            u32x4 w;
            w.x = __builtin_amdgcn_perm(reg[1][d], reg[0][d], sel);
            w.y = __builtin_amdgcn_perm(reg[3][d], reg[2][d], sel);
            w.z = __builtin_amdgcn_perm(reg[5][d], reg[4][d], sel);
            w.w = __builtin_amdgcn_perm(reg[7][d], reg[6][d], sel);
            *reinterpret_cast<u32x4*>(base + img_off((rg << 3) + mm, kg)) = w;
        }
    }
};

// one slot of the shadow path's register ring
template <int BM, int BN, int NT>
struct SlotKK {
    Stage16<GD_LAY_KC, BM, NT> a;
    Stage16<GD_LAY_KC, BN, NT> b;
    static constexpr int LOADS = Stage16<GD_LAY_KC, BM, NT>::LOADS + Stage16<GD_LAY_KC, BN, NT>::LOADS;
    __device__ __forceinline__ void load_asm(const GdGemm& g, int m0, int n0, int k0, int tid) {
        a.load_asm(static_cast<const unsigned short*>(g.A16), g.lda16, m0, g.M, k0, tid);
        b.load_asm(static_cast<const unsigned short*>(g.B16), g.ldb16, n0, g.N, k0, tid);
    }
    __device__ __forceinline__ void pin() {
        a.pin();
        b.pin();
    }
    __device__ __forceinline__ void store(char* img, int tid) const {
        a.store(img, tid);
        b.store(img + BM * 128, tid);
    }
};

template <int LAYA, int LAYB, int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool S16>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void gemm_bf16_kernel(const GdGemm g) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    static_assert(TM * 16 * WAVES_M == BM && TN * 16 * WAVES_N == BN, "tile must split into 16x16 blocks");
    constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
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
    const int nt = (g.dbg & 1) ? 0 : (kend - kbeg + BK - 1) / BK;  // (dbg: timing ablations, GDMCF_BF16_DBG; 0 in production)

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const char* cur) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const bf16x8*>(cur + img_off(wm0 + 16 * i + r, 4 * c + q));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const bf16x8*>(cur + A_BYTES + img_off(wn0 + 16 * j + r, 4 * c + q));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };
    constexpr bool RING_KK = S16 && LAYA == GD_LAY_KC && LAYB == GD_LAY_KC;
    constexpr bool RING_MM = S16 && LAYA == GD_LAY_MC && LAYB == GD_LAY_MC && BM + BN <= NT;
    if constexpr (RING_KK || RING_MM) {
        // two register slots: tiles it+1 and it+2 are in flight while tile it is multiplied (see gload16_u).  Measured on
        // the Amazon-Book shape: forward 0.128 -> 0.115 ms, fused-loss forward 0.246 -> 0.233 ms, weight gradients (paired
        // units, Stage16Pair) 0.269 -> 0.203 ms; Yelp shape weight gradients 0.187 -> 0.080 ms.  The K-contiguous x
        // row-contiguous product (dh) keeps the single-slot loop below: with two slots of 48 registers the 208x256 kernel
        // spills (0.132 -> 0.160 ms).
        using Slot = typename std::conditional<RING_KK, SlotKK<BM, BN, NT>, Stage16Pair<BM, BN, NT>>::type;
        Slot s0, s1;
        constexpr int LPT = Slot::LOADS;
        static_assert(LPT <= 63, "vmcnt is a 6-bit counter");
#define GD_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
        char* const L0 = lds;
        char* const L1 = lds + STAGE_BYTES;
        if (nt > 0) {
            s0.load_asm(g, m0, n0, kbeg, tid);
            if (nt > 1) {
                s1.load_asm(g, m0, n0, kbeg + BK, tid);
                GD_WAIT_VM(LPT);
            } else {
                GD_WAIT_VM(0);
            }
            s0.pin();
            s0.store(L0, tid);
        }
        __syncthreads();
        for (int it = 0; it < nt; it += 2) {
            // even tile `it` in L0; slot 1 holds tile it+1 (in flight); slot 0 is free
            const bool ld2 = it + 2 < nt;
            if (ld2) s0.load_asm(g, m0, n0, kbeg + (it + 2) * BK, tid);
            compute(L0);
            if (it + 1 < nt) {
                if (ld2) GD_WAIT_VM(LPT); else GD_WAIT_VM(0);
                s1.pin();
                s1.store(L1, tid);
            }
            __syncthreads();
            if (it + 1 < nt) {
                const bool ld3 = it + 3 < nt;
                if (ld3) s1.load_asm(g, m0, n0, kbeg + (it + 3) * BK, tid);
                compute(L1);
                if (ld2) {
                    if (ld3) GD_WAIT_VM(LPT); else GD_WAIT_VM(0);
                    s0.pin();
                    s0.store(L0, tid);
                }
                __syncthreads();
            }
        }
        GD_WAIT_VM(0);
#undef GD_WAIT_VM
    } else {
        typename std::conditional<S16, Stage16<LAYA, BM, NT>, Stage<LAYA, BM, NT>>::type sa;
        typename std::conditional<S16, Stage16<LAYB, BN, NT>, Stage<LAYB, BN, NT>>::type sb;
        auto load_tile = [&](int k0) {
            if constexpr (S16) {
                sa.load(static_cast<const unsigned short*>(g.A16), g.lda16, m0, g.M, k0, tid);
                sb.load(static_cast<const unsigned short*>(g.B16), g.ldb16, n0, g.N, k0, tid);
            } else {
                stage_load(sa, g.A, g.lda, m0, g.M, k0, kend, tid);
                stage_load(sb, g.B, g.ldb, n0, g.N, k0, kend, tid);
            }
        };
        auto store_tile = [&](char* img, int k0) {
            if constexpr (S16) {
                sa.store(img, tid);
                sb.store(img + A_BYTES, tid);
            } else {
                sa.store(img, k0, kend, tid);
                sb.store(img + A_BYTES, k0, kend, tid);
            }
        };
        if (nt > 0) {
            load_tile(kbeg);
            store_tile(lds, kbeg);
        }
        __syncthreads();
        for (int it = 0; it < nt; ++it) {
            const char* cur = lds + (it & 1) * STAGE_BYTES;
            char* nxt = lds + ((it + 1) & 1) * STAGE_BYTES;
            const bool more = it + 1 < nt;
            if (more) load_tile(kbeg + (it + 1) * BK);
            compute(cur);
            if (more) store_tile(nxt, kbeg + (it + 1) * BK);
            __syncthreads();
        }
    }
    if (g.dbg & 2) return;
    if constexpr (EPI == GD_EPI_ADAMW || (EPI == GD_EPI_STORE && S16))
        gemm_epilogue_rows<BM, BN, TM, TN, WAVES_M, WAVES_N, EPI, 64 * (BM + BN), NT>(acc, g, m0, n0, wn0, r, q, wave, tid,
                                                                                     smem);
    else
        gemm_epilogue<BM, TM, TN, WAVES_N, EPI>(acc, g, m0, n0, wm0, wn0, r, q, split, tile_n, wave, tid, smem);
}

template <int LAYA, int LAYB, int BM, int BN, int WM, int WN, int EPI, bool S16>
int launch_one(GdGemm& g, hipStream_t s) {
    constexpr size_t lds = 2 * (size_t)(BM + BN) * 128;
    auto kern = gemm_bf16_kernel<LAYA, LAYB, BM, BN, WM, WN, EPI, S16>;
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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
    return gd_launch_status("gemm_bf16");
}

template <int LAYA, int LAYB, int EPI>
int launch_class(int cls, GdGemm& g, hipStream_t s) {
    // operands come from their bf16 shadows when both have one
    const bool s16 = g.A16 != nullptr && g.B16 != nullptr;
    switch (cls) {
        case 0: return s16 ? launch_one<LAYA, LAYB, 80, 128, 1, 4, EPI, true>(g, s)
                           : launch_one<LAYA, LAYB, 80, 128, 1, 4, EPI, false>(g, s);
        case 1: return s16 ? launch_one<LAYA, LAYB, 128, 128, 2, 2, EPI, true>(g, s)
                           : launch_one<LAYA, LAYB, 128, 128, 2, 2, EPI, false>(g, s);
        case 2: return s16 ? launch_one<LAYA, LAYB, 64, 64, 2, 2, EPI, true>(g, s)
                           : launch_one<LAYA, LAYB, 64, 64, 2, 2, EPI, false>(g, s);
        case 3: return s16 ? launch_one<LAYA, LAYB, 208, 256, 1, 8, EPI, true>(g, s)
                           : launch_one<LAYA, LAYB, 208, 256, 1, 8, EPI, false>(g, s);
    }
    gdmcf_set_error("bad gemm shape class %d", cls);
    return GDMCF_E_ARG;
}

}  // namespace

int gd_gemm_bf16_launch(int layA, int layB, int epi, int cls, GdGemm& g, hipStream_t s) {
    // timing ablations of the weight-gradient products (tools/bf16_fused_sweep.sh): bit 0 skips the k loop, bit 1 the epilogue
    static const int dbg = getenv("GDMCF_BF16_DBG") ? atoi(getenv("GDMCF_BF16_DBG")) : 0;
    if (dbg && layA == GD_LAY_MC && layB == GD_LAY_MC) g.dbg = dbg;
    if (g.kchunk > 0 && g.kchunk % BK != 0) {
        gdmcf_set_error("gemm_bf16: kchunk %d is not a multiple of %d", g.kchunk, BK);
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
    gdmcf_set_error("unsupported bf16 gemm variant (layA=%d layB=%d epi=%d)", layA, layB, epi);
    return GDMCF_E_UNSUPPORTED;
}
