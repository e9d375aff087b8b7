// f32 GEMM core on the CDNA4 f32-input matrix instruction v_mfma_f32_16x16x4_f32.
//
// One templated kernel serves every dense product of the denoiser (reference models/DNN.py:79-86
// forward, autograd backward at main.py:350):
//   KC/KC  C = A[M,K] * B[N,K]^T          forward layers (both operands K-contiguous)
//   KC/MC  C = A[M,K] * B[K,N]            grad wrt layer input  (dZ @ W)
//   MC/MC  C = A[K,M]^T * B[K,N]          grad wrt weight       (dZ^T @ A)
// Tiles are staged global -> registers -> LDS (double buffered, loads for tile t+1 issued before
// the MFMAs of tile t).  LDS images: K-contiguous operands as [row][BK+4] read with one
// ds_read_b128 per 16-deep k chunk; row-contiguous operands as [k][rows+pad] read with
// ds_read_b32.  The k index inside a 16-chunk is permuted (lane group q, step s -> k = 4q+s)
// identically for A and B so a b128 read feeds four consecutive MFMAs.
// The exact f32 MFMA is a k-ordered fmaf chain, so results are deterministic for a given
// (tile, split) configuration.
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"

#ifdef GD_PROBE_NO_LDSWRITE
#define GD_PROBE_LDSWRITE_OFF 1
#else
#define GD_PROBE_LDSWRITE_OFF 0
#endif

namespace {

constexpr int NTHREADS = 256;

// LDS images (bank rules: MI355X_MICROARCH.md, LDS):
//  * K-contiguous operand: rows of BK floats with NO padding; the 16-byte slot s of row r is stored at slot
//    s ^ ((r >> 1) & 7).  ds_read_b128 serves a wave in four 16-lane groups, each holding all 16 rows of an
//    MFMA block at two neighbouring k-slots: with this swizzle the 16 addresses fall on 16 distinct
//    16-byte slots of the 256-byte bank row (the former [row][BK+4] padding was 2-way conflicted).
//  * row-contiguous operand: [k][rows + 4]; lane group q reads k-row 4q+s, so rows 4 apart must differ by
//    16 banks: (rows+4) mod 8 == 4.
template <int LAY, int R, int BK>
struct TileGeom {
    static_assert(LAY != GD_LAY_KC || BK == 32, "the XOR swizzle assumes 8 slots (32 floats) per row");
    static_assert(R % 8 == 0, "tile rows must be a multiple of 8");
    static constexpr int LD = (LAY == GD_LAY_KC) ? BK : (R + 4);
    static constexpr int FLOATS = (LAY == GD_LAY_KC) ? R * LD : BK * LD;
    static constexpr int F4 = R * BK / 4;
    static constexpr int NL = (F4 + NTHREADS - 1) / NTHREADS;
    // float offset of 16-byte slot `slot` (0..7) of row r in a K-contiguous image
    __device__ static __forceinline__ int kc_off(int r, int slot) { return r * BK + ((slot ^ ((r >> 1) & 7)) << 2); }
};

// ---- global -> register staging -----------------------------------------------------------
// 16-byte vectors that are only 4-byte aligned: gfx950 runs with unaligned access enabled and
// hipcc emits global_load_dwordx4 for them, so odd row strides (nn.Linear in_features = 34405)
// still stream with the widest load.
typedef f32x4 f32x4_u __attribute__((aligned(4)));

// Tile loads are issued as inline asm so that hipcc neither counts nor waits for them: the kernel keeps
// two tiles in flight and places the counted s_waitcnt vmcnt(N) itself (stage_wait below).  Left to the
// compiler, the waits degenerate to vmcnt(0..4) at joins/back-edges and drain the second stage.
__device__ __forceinline__ f32x4 gload16(const float* p) {
    f32x4 v;
#ifdef GD_PROBE_NO_GLOBAL  // tools/gemm_probe.hip ablation only
    v = f32x4{1.f, 1.f, 1.f, 1.f};
    asm volatile("" : "+v"(v) : "v"(p));
#else
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
#endif
    return v;
}

template <int LAY, int R, int BK>
struct TileStage {
    using G = TileGeom<LAY, R, BK>;
    f32x4 reg[G::NL];

    // Interior tile: k0 + BK <= kend and (MC layout) row0 + R <= nrows or nrows % 4 == 0.  Nothing but address arithmetic and
    // loads, so all of them issue back to back and the only wait is at the LDS write after the MFMAs.
    // KC rows past `nrows` are clamped onto valid rows (their products land in accumulator rows/columns
    // that the epilogue never stores).
    __device__ __forceinline__ void load_plain(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0,
                                               int tid) {
#pragma unroll
        for (int i = 0; i < G::NL; ++i) {
            int idx = tid + i * NTHREADS;
            if (G::NL * NTHREADS != G::F4) idx = min(idx, G::F4 - 1);
            if (LAY == GD_LAY_KC) {
                const int r = idx / (BK / 4), kk = (idx % (BK / 4)) * 4;
                const int gr = min(row0 + r, nrows - 1);
                reg[i] = gload16(P + (int64_t)gr * ld + (k0 + kk));
            } else {
                // partial last row block with nrows % 4 == 0 (see load()): groups past the edge are clamped onto the
                // last valid one; their products land in accumulator rows/columns that are never stored
                const int kk = idx / (R / 4), r = (idx % (R / 4)) * 4;
                reg[i] = gload16(P + (int64_t)(k0 + kk) * ld + min(row0 + r, nrows - 4));
            }
        }
    }

    // Edge tile (K tail and/or last row block), still branch-free: addresses are clamped onto valid
    // elements here; the in-register shift of a vector straddling the edge and the zero fill of k >= kend
    // are deferred to store_edge(), i.e. after the MFMAs, so these loads stay in flight like plain ones.
    // No load leaves [0,nrows) x [0,kend).  Needs kend >= 4 (KC) / nrows >= 4 (MC).
    __device__ __forceinline__ void load_edge(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0,
                                              int kend, int tid) {
#pragma unroll
        for (int i = 0; i < G::NL; ++i) {
            int idx = tid + i * NTHREADS;
            if (G::NL * NTHREADS != G::F4) idx = min(idx, G::F4 - 1);
            if (LAY == GD_LAY_KC) {
                const int r = idx / (BK / 4), kk = (idx % (BK / 4)) * 4;
                const int gr = min(row0 + r, nrows - 1);
                const int gc = max(min(k0 + kk, kend - 4), 0);
                reg[i] = gload16(P + (int64_t)gr * ld + gc);
            } else {
                const int kk = idx / (R / 4), r = (idx % (R / 4)) * 4;
                const int gc = min(row0 + r, nrows - 4);
                const int gk = min(k0 + kk, kend - 1);
                reg[i] = gload16(P + (int64_t)gk * ld + gc);
            }
        }
    }

    __device__ __forceinline__ void fix_edge(int row0, int nrows, int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < G::NL; ++i) {
            int idx = tid + i * NTHREADS;
            if (G::NL * NTHREADS != G::F4) idx = min(idx, G::F4 - 1);
            f32x4 v = reg[i];
            if (LAY == GD_LAY_KC) {
                const int gk = k0 + (idx % (BK / 4)) * 4;
                const int sh = gk - max(min(gk, kend - 4), 0);  // 0 interior, 1..3 straddling kend, >= 4 beyond
                v.x = sh == 0 ? v.x : (sh == 1 ? v.y : (sh == 2 ? v.z : v.w));
                v.y = sh == 0 ? v.y : (sh == 1 ? v.z : v.w);
                v.z = sh == 0 ? v.z : v.w;
                v.x = (gk + 0 < kend) ? v.x : 0.f;
                v.y = (gk + 1 < kend) ? v.y : 0.f;
                v.z = (gk + 2 < kend) ? v.z : 0.f;
                v.w = (gk + 3 < kend) ? v.w : 0.f;
            } else {
                const int kk = idx / (R / 4), gr = row0 + (idx % (R / 4)) * 4;
                const int sh = gr - min(gr, nrows - 4);
                v.x = sh == 0 ? v.x : (sh == 1 ? v.y : (sh == 2 ? v.z : v.w));
                v.y = sh == 0 ? v.y : (sh == 1 ? v.z : v.w);
                v.z = sh == 0 ? v.z : v.w;
                const bool kz = (k0 + kk >= kend);
                v.x = kz ? 0.f : v.x;
                v.y = kz ? 0.f : v.y;
                v.z = kz ? 0.f : v.z;
                v.w = kz ? 0.f : v.w;
            }
            reg[i] = v;
        }
    }

    int mode, m_row0, m_nrows, m_k0, m_kend;  // how this stage was loaded (wave-uniform), for store()

    // --- per-slot interface of the hand-interleaved steady-state loop (interior tiles only) ---
    // source address of slot i for the K tile starting at k0 (rows of K-contiguous operands clamped)
    static __device__ __forceinline__ const float* slot_ptr(const float* __restrict__ P, int64_t ld, int row0,
                                                            int nrows, int k0, int i, int tid) {
        int idx = tid + i * NTHREADS;
        if (G::NL * NTHREADS != G::F4) idx = min(idx, G::F4 - 1);
        if (LAY == GD_LAY_KC) {
            const int r = idx / (BK / 4), kk = (idx % (BK / 4)) * 4;
            return P + (int64_t)min(row0 + r, nrows - 1) * ld + (k0 + kk);
        }
        // M-contiguous operand, last row block: 4-float groups past `nrows` are clamped onto the last valid group
        // (callers take this path for partial blocks only when nrows % 4 == 0, so no group straddles the edge);
        // what they carry lands in accumulator rows/columns the epilogue never stores.
        const int kk = idx / (R / 4), r = (idx % (R / 4)) * 4;
        return P + (int64_t)(k0 + kk) * ld + min(row0 + r, nrows - 4);
    }
    static __device__ __forceinline__ int64_t tile_step(int64_t ld) { return (LAY == GD_LAY_KC) ? BK : (int64_t)BK * ld; }
    // LDS float offset of slot i (-1: this thread has no element in the partial last slot)
    static __device__ __forceinline__ int slot_lds(int i, int tid) {
        const int idx = tid + i * NTHREADS;
        if (G::NL * NTHREADS != G::F4 && idx >= G::F4) return -1;
        if (LAY == GD_LAY_KC) return G::kc_off(idx / (BK / 4), idx % (BK / 4));
        return (idx / (R / 4)) * G::LD + (idx % (R / 4)) * 4;
    }

    // Pins the stage registers behind a wait the caller has just executed (the asm is empty; the "+v"
    // operands make every register of the stage opaque at this point, guide 5.7 form ii).
    __device__ __forceinline__ void pin() {
#pragma unroll
        for (int i = 0; i < G::NL; ++i) asm volatile("" : "+v"(reg[i]));
    }

    __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0,
                                         int kend, int kbeg, bool rows_full, int tid) {
        // exactly two variants with the same number of loads in the same order, so the compiler's counted
        // vmcnt waits stay exact across the join (a third, predicated path would force conservative waits
        // that drain the second register stage).  Degenerate shapes never get here (gemm_small.hip takes them).
        const bool k_full = (k0 + BK <= kend);
        if (k_full && (LAY == GD_LAY_KC || rows_full || (nrows & 3) == 0)) {
            mode = 0;
            load_plain(P, ld, row0, nrows, k0, tid);
        } else {
            mode = 1;
            m_row0 = row0; m_nrows = nrows; m_k0 = k0; m_kend = kend;
            load_edge(P, ld, row0, nrows, k0, kend, tid);
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ lds, int tid) {
        if (mode == 1) fix_edge(m_row0, m_nrows, m_k0, m_kend, tid);
#pragma unroll
        for (int i = 0; i < G::NL; ++i) {
            const int idx = tid + i * NTHREADS;
            if (G::NL * NTHREADS == G::F4 || idx < G::F4) {
                if (LAY == GD_LAY_KC) {
                    const int r = idx / (BK / 4), slot = idx % (BK / 4);
                    *reinterpret_cast<f32x4*>(&lds[G::kc_off(r, slot)]) = reg[i];
                } else {
                    const int kk = idx / (R / 4), r = (idx % (R / 4)) * 4;
                    *reinterpret_cast<f32x4*>(&lds[kk * G::LD + r]) = reg[i];
                }
            }
        }
    }
};

// ---- LDS -> MFMA operand fragments -----------------------------------------------------------
template <int LAY, int R, int BK, int T>
__device__ __forceinline__ void load_frag(const float* __restrict__ lds, int row_base, int c, int r, int q,
                                          float (&f)[T][4]) {
    using G = TileGeom<LAY, R, BK>;
#ifdef GD_PROBE_NO_LDSREAD  // ablation (tools/gemm_probe.hip): operands stay whatever the registers hold
    for (int t = 0; t < T; ++t) for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(f[t][s]));
    return;
#endif
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (LAY == GD_LAY_KC) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[G::kc_off(row_base + t * 16 + r, 4 * c + q)]);
            f[t][0] = v.x;
            f[t][1] = v.y;
            f[t][2] = v.z;
            f[t][3] = v.w;
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) f[t][s] = lds[(16 * c + 4 * q + s) * G::LD + row_base + t * 16 + r];
        }
    }
}


template <int LAYA, int LAYB, int BM, int BN, int BK, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(const GdGemm g) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    static_assert(TM * 16 * WAVES_M == BM && TN * 16 * WAVES_N == BN, "tile must split into 16x16 blocks");
    using GA = TileGeom<LAYA, BM, BK>;
    using GB = TileGeom<LAYB, BN, BK>;
    constexpr int STAGE_FLOATS = GA::FLOATS + GB::FLOATS;

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WTM;
    const int wn0 = (wave % WAVES_N) * WTN;
    const int r = lane & 15, q = lane >> 4;

    // XCD-aware bijective remap: consecutive logical ids run on one XCD (blocks b and b+8 share an
    // XCD under round-robin dispatch), so tiles that share an operand panel hit the same L2.
    const int nwg = gridDim.x;
    const int id = blockIdx.x;
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

    // Two register stages: the loads of tile t+2 are issued before the MFMAs of tile t, the registers of
    // tile t+1 (issued one iteration earlier) are written to LDS after them.  Every load therefore has
    // two full iterations to land -- with ~4 us loaded L2/HBM latency a CU needs ~100 KB in flight to keep
    // its MFMA pipes busy, one tile ahead only gives half of that (measured: 40-60 % MFMA busy).
    TileStage<LAYA, BM, BK> sa0, sa1;
    TileStage<LAYB, BN, BK> sb0, sb1;
    const bool a_full = (m0 + BM <= g.M), b_full = (n0 + BN <= g.N);
    float* const L0 = smem;
    float* const L1 = smem + STAGE_FLOATS;

    auto compute = [&](const float* As) {
        const float* Bs = As + GA::FLOATS;
#pragma unroll
        for (int c = 0; c < BK / 16; ++c) {
            float fa[TM][4], fb[TN][4];
#ifdef GD_PROBE_NO_LDSREAD
            // ablation: operands stay whatever they were (uninitialised registers), no LDS traffic at all
            for (int i = 0; i < TM; ++i) for (int s = 0; s < 4; ++s) asm volatile("" : "=v"(fa[i][s]));
            for (int j = 0; j < TN; ++j) for (int s = 0; s < 4; ++s) asm volatile("" : "=v"(fb[j][s]));
#else
            load_frag<LAYA, BM, BK, TM>(As, wm0, c, r, q, fa);
            load_frag<LAYB, BN, BK, TN>(Bs, wn0, c, r, q, fb);
#endif
#ifdef GD_PROBE_NO_MFMA
            for (int s = 0; s < 4; ++s) for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j][s] += fa[i][s] + fb[j][s];
#else
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
#endif
        }
    };

    constexpr int LOADS_PER_TILE = GA::NL + GB::NL;
#define GD_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#ifdef GD_PROBE_NO_BARRIER
#define __syncthreads() ((void)0)
#define GD_RAW_BARRIER() ((void)0)
#else
#define GD_RAW_BARRIER() __builtin_amdgcn_s_barrier()
#endif
    if (nt > 0) {
        sa0.load(g.A, g.lda, m0, g.M, kbeg, kend, kbeg, a_full, tid);
        sb0.load(g.B, g.ldb, n0, g.N, kbeg, kend, kbeg, b_full, tid);
        if (nt > 1) {
            sa1.load(g.A, g.lda, m0, g.M, kbeg + BK, kend, kbeg, a_full, tid);
            sb1.load(g.B, g.ldb, n0, g.N, kbeg + BK, kend, kbeg, b_full, tid);
            GD_WAIT_VM(LOADS_PER_TILE);  // tile 0 landed, tile 1 still in flight
        } else {
            GD_WAIT_VM(0);
        }
        sa0.pin();
        sb0.pin();
        sa0.store(L0, tid);
        sb0.store(L0 + GA::FLOATS, tid);
    }
    __syncthreads();

    int it = 0;
    // ---- hand-interleaved steady state -----------------------------------------------------------------
    // Measured (tools/gemm_probe): with the loads, the counted wait and the LDS writes sitting in one block
    // after 80 back-to-back MFMAs, time = MFMA time + everything-else time; the second resident wave hides
    // little of it.  Here the same work is threaded through the MFMA stream of the wave itself: the 7 loads of
    // tile it+2 ride in the first half (k-chunk 0), the 7 LDS writes of tile it+1 in the second half, the
    // fragments of chunk 1 are fetched during chunk 0.  sched_barrier(0) pins the order.  Interior tiles
    // only; the generic loop below finishes the tail (and runs boundary workgroups entirely).
    const int nt_full = (kend - kbeg) / BK;
    // (an M-contiguous operand's partial last block qualifies when its extent is a multiple of 4: slot_ptr clamps)
    if ((LAYA == GD_LAY_KC || a_full || (g.M & 3) == 0) && (LAYB == GD_LAY_KC || b_full || (g.N & 3) == 0) && nt_full >= 4) {
        const float* pa[GA::NL];
        const float* pb[GB::NL];
        int wa[GA::NL], wb[GB::NL];
#pragma unroll
        for (int i = 0; i < GA::NL; ++i) {
            pa[i] = TileStage<LAYA, BM, BK>::slot_ptr(g.A, g.lda, m0, g.M, kbeg + 2 * BK, i, tid);
            wa[i] = TileStage<LAYA, BM, BK>::slot_lds(i, tid);
        }
#pragma unroll
        for (int i = 0; i < GB::NL; ++i) {
            pb[i] = TileStage<LAYB, BN, BK>::slot_ptr(g.B, g.ldb, n0, g.N, kbeg + 2 * BK, i, tid);
            wb[i] = TileStage<LAYB, BN, BK>::slot_lds(i, tid);
        }
        const int64_t stepa = TileStage<LAYA, BM, BK>::tile_step(g.lda), stepb = TileStage<LAYB, BN, BK>::tile_step(g.ldb);
        constexpr int NLT = GA::NL + GB::NL;  // loads (and LDS writes) per tile and thread
        constexpr int NCH = BK / 16;          // k-chunks per tile, 4 MFMA groups each
        constexpr int NG = 4 * NCH;           // MFMA groups per tile
        constexpr int HG = NG / 2;            // loads ride in groups [0,HG), LDS writes in [HG,NG-1)
        constexpr int PERL = (NLT + HG - 1) / HG;
        constexpr int WG_ = NG - 1 - HG;      // groups that carry LDS writes
        constexpr int PERW = (NLT + WG_ - 1) / WG_;
        static_assert(WG_ >= 1, "need at least one group for the LDS writes");

        // Fragment registers live across half-iterations: chunk c of a tile uses parity (P0 + c) & 1 and the
        // first fragments of the NEXT tile are fetched (after the barrier) under the last MFMA group.
        float fra[2][TM][4], frb[2][TN][4];
        load_frag<LAYA, BM, BK, TM>(L0, wm0, 0, r, q, fra[0]);
        load_frag<LAYB, BN, BK, TN>(L0 + GA::FLOATS, wn0, 0, r, q, frb[0]);

        // one half-iteration: compute tile from Lc, load next-next tile into (la, lb), write (sa_, sb_) to Ln.
        // All LDS reads of Lc are issued by group 1 of the last chunk, all writes to Ln by group NG-2: ONE barrier
        // before the last group covers both hazards, and the last group's MFMAs hide the barrier skew and the
        // latency of the next tile's first fragment reads.
#define GD_HALF(Lc, Ln, la, lb, sa_, sb_, P0)                                                                      \
        {                                                                                                          \
            _Pragma("unroll") for (int gi = 0; gi < NG; ++gi) {                                                    \
                const int ch = gi >> 2, sgrp = gi & 3, par = ((P0) + ch) & 1;                                      \
                if (gi == HG) {                                                                                    \
                    GD_WAIT_VM(NLT); /* the tile loaded one half-iteration ago has landed; the new one flies */    \
                    sa_.pin();                                                                                     \
                    sb_.pin();                                                                                     \
                }                                                                                                  \
                if (gi == NG - 1) {                                                                                \
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
                    GD_RAW_BARRIER();                                                                              \
                    load_frag<LAYA, BM, BK, TM>(Ln, wm0, 0, r, q, fra[((P0) + NCH) & 1]);                          \
                    load_frag<LAYB, BN, BK, TN>(Ln + GA::FLOATS, wn0, 0, r, q, frb[((P0) + NCH) & 1]);             \
                }                                                                                                  \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                     \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fra[par][i][sgrp], frb[par][j][sgrp], acc[i][j], 0, 0, 0); \
                if (gi < HG) {                                                                                     \
                    _Pragma("unroll") for (int u = gi * PERL; u < (gi + 1) * PERL && u < NLT; ++u) {                \
                        if (u < GA::NL) { la.reg[u] = gload16(pa[u]); pa[u] += stepa; }                             \
                        else { lb.reg[u - GA::NL] = gload16(pb[u - GA::NL]); pb[u - GA::NL] += stepb; }             \
                    }                                                                                              \
                } else if (gi < NG - 1) {                                                                          \
                    _Pragma("unroll") for (int u = (gi - HG) * PERW; u < (gi - HG + 1) * PERW && u < NLT; ++u) {    \
                        if (GD_PROBE_LDSWRITE_OFF) continue;                                                        \
                        if (u < GA::NL) { if (wa[u] >= 0) *reinterpret_cast<f32x4*>(&(Ln)[wa[u]]) = sa_.reg[u]; }   \
                        else if (wb[u - GA::NL] >= 0)                                                               \
                            *reinterpret_cast<f32x4*>(&(Ln)[GA::FLOATS + wb[u - GA::NL]]) = sb_.reg[u - GA::NL];    \
                    }                                                                                              \
                }                                                                                                  \
                if (sgrp == 1 && ch + 1 < NCH) { /* fetch the next chunk's fragments under this chunk's MFMAs */  \
                    load_frag<LAYA, BM, BK, TM>(Lc, wm0, ch + 1, r, q, fra[par ^ 1]);                              \
                    load_frag<LAYB, BN, BK, TN>(Lc + GA::FLOATS, wn0, ch + 1, r, q, frb[par ^ 1]);                 \
                }                                                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
            }                                                                                                      \
        }
        // invariant at the top (even `it`): L0 holds tile it (its chunk-0 fragments already in registers),
        // stage-1 registers hold tile it+1 (in flight), stage-0 registers are free; tiles it+2, it+3 interior.
        for (; it + 3 < nt_full; it += 2) {
            GD_HALF(L0, L1, sa0, sb0, sa1, sb1, 0);          // tile it;   loads it+2 -> stage 0, writes it+1 -> L1
            GD_HALF(L1, L0, sa1, sb1, sa0, sb0, (NCH & 1));  // tile it+1; loads it+3 -> stage 1, writes it+2 -> L0
        }
#undef GD_HALF
        // (the fragments prefetched for tile `it` are simply re-read by the generic loop below)
        __syncthreads();
        sa0.mode = sb0.mode = sa1.mode = sb1.mode = 0;
    }

    for (; it < nt; it += 2) {
        // even tile `it`: LDS stage 0; stage-1 registers hold tile it+1 (in flight); stage-0 registers are free
        const bool ld2 = (it + 2 < nt);
        if (ld2) {
            sa0.load(g.A, g.lda, m0, g.M, kbeg + (it + 2) * BK, kend, kbeg, a_full, tid);
            sb0.load(g.B, g.ldb, n0, g.N, kbeg + (it + 2) * BK, kend, kbeg, b_full, tid);
        }
        compute(L0);
        if (it + 1 < nt) {
            if (ld2) GD_WAIT_VM(LOADS_PER_TILE); else GD_WAIT_VM(0);  // tile it+1 landed
            sa1.pin();
            sb1.pin();
            sa1.store(L1, tid);
            sb1.store(L1 + GA::FLOATS, tid);
        }
        __syncthreads();
        if (it + 1 < nt) {
            // odd tile `it+1`: LDS stage 1; stage-0 registers hold tile it+2 (in flight)
            const bool ld3 = (it + 3 < nt);
            if (ld3) {
                sa1.load(g.A, g.lda, m0, g.M, kbeg + (it + 3) * BK, kend, kbeg, a_full, tid);
                sb1.load(g.B, g.ldb, n0, g.N, kbeg + (it + 3) * BK, kend, kbeg, b_full, tid);
            }
            compute(L1);
            if (ld2) {
                if (ld3) GD_WAIT_VM(LOADS_PER_TILE); else GD_WAIT_VM(0);  // tile it+2 landed
                sa0.pin();
                sb0.pin();
                sa0.store(L0, tid);
                sb0.store(L0 + GA::FLOATS, tid);
            }
            __syncthreads();
        }
    }
    GD_WAIT_VM(0);
#undef GD_WAIT_VM
#ifdef GD_PROBE_NO_BARRIER
#undef __syncthreads
#endif

    gemm_epilogue<BM, TM, TN, WAVES_N, EPI>(acc, g, m0, n0, wm0, wn0, r, q, split, tile_n, wave, tid, smem);
}

// dynamic LDS of the wave-specialised kernel in floats: the two operand stages, or -- for the fused-AdamW product,
// whose result goes back through LDS as full rows (gemm_epilogue_rows) -- the whole result tile when that still
// leaves room for two workgroups per CU
template <int LAYA, int LAYB, int BM, int BN, int BK, int WAVES_M, int EPI>
constexpr int spec_lds_floats() {
    constexpr int main_f = 2 * (TileGeom<LAYA, BM, BK>::FLOATS + TileGeom<LAYB, BN, BK>::FLOATS);
    constexpr int tile_f = BM * (BN + 4);
    constexpr bool rows = (EPI == GD_EPI_ADAMW);
    return (rows && tile_f > main_f && tile_f * 4 <= 78 * 1024) ? tile_f : main_f;
}

// ---- wave-specialised variant ------------------------------------------------------------------------------
// 512 threads: waves 0-3 issue nothing but LDS fragment reads and MFMAs, waves 4-7 do all global loads and LDS
// writes (same two-tile-ahead register staging, same LDS images, same edge handling).  Motivation (measured):
// a global_load_dwordx4 costs the issuing wave ~60 cycles of issue time, 7 per tile = the ~415-cycle per-tile
// overhead of the all-in-one kernel; in a loader wave that time is off the MFMA waves' critical path.
// One s_barrier per tile for all eight waves: after it tile t+1 is complete in LDS and tile t's stage is free.
template <int LAYA, int LAYB, int BM, int BN, int BK, int WAVES_M, int WAVES_N, int EPI, int NSTG>
__global__ __launch_bounds__(2 * NTHREADS, 4) void gemm_f32_spec_kernel(const GdGemm g) {
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    using GA = TileGeom<LAYA, BM, BK>;
    using GB = TileGeom<LAYB, BN, BK>;
    constexpr int STAGE_FLOATS = GA::FLOATS + GB::FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const bool loader = threadIdx.x >= NTHREADS;
    const int tid = threadIdx.x & (NTHREADS - 1);
    const int lane = tid & 63, wave = tid >> 6;
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
    float* const L0 = smem;
    float* const L1 = smem + STAGE_FLOATS;
    constexpr int LOADS_PER_TILE = GA::NL + GB::NL;
#define GD_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

    if (loader) {
        // NSTG register stages: the loads of tiles it+1 .. it+NSTG-1 are in flight while tile `it` is computed
        // (LDS stays double buffered).  Loader waves hold no accumulators, so depth costs only idle VGPRs.
        TileStage<LAYA, BM, BK> sa[NSTG];
        TileStage<LAYB, BN, BK> sb[NSTG];
        const bool a_full = (m0 + BM <= g.M), b_full = (n0 + BN <= g.N);
        // wait until every load but those of the `after` youngest tiles has landed (workgroup-uniform switch)
        auto wait_tiles = [](int after) {
            constexpr int LPT = GA::NL + GB::NL;
            switch (after) {
                case 0: GD_WAIT_VM(0); break;
                case 1: GD_WAIT_VM(LPT); break;
                case 2: GD_WAIT_VM(2 * LPT); break;
                case 3: GD_WAIT_VM(3 * LPT); break;
                case 4: GD_WAIT_VM(4 * LPT); break;
                case 5: GD_WAIT_VM(5 * LPT); break;
                default: GD_WAIT_VM(6 * LPT); break;
            }
        };
        static_assert(NSTG >= 2 && NSTG <= 8 && (NSTG - 2) * LOADS_PER_TILE <= 63, "vmcnt is a 6-bit counter");
#pragma unroll
        for (int u = 0; u < NSTG - 1; ++u) {
            if (u < nt) {
                sa[u].load(g.A, g.lda, m0, g.M, kbeg + u * BK, kend, kbeg, a_full, tid);
                sb[u].load(g.B, g.ldb, n0, g.N, kbeg + u * BK, kend, kbeg, b_full, tid);
            }
        }
        if (nt > 0) {
            wait_tiles(min(NSTG - 2, nt - 1));
            sa[0].pin();
            sb[0].pin();
            sa[0].store(L0, tid);
            sb[0].store(L0 + GA::FLOATS, tid);
        }
        __syncthreads();
        for (int base = 0; base < nt; base += NSTG) {
#pragma unroll
            for (int u = 0; u < NSTG; ++u) {
                const int it = base + u;  // MFMA waves compute tile `it`; this step makes tile it+1 visible in LDS
                if (it >= nt) break;
                const int s_new = (u + NSTG - 1) % NSTG, s_nxt = (u + 1) % NSTG;
                if (it + NSTG - 1 < nt) {
                    sa[s_new].load(g.A, g.lda, m0, g.M, kbeg + (it + NSTG - 1) * BK, kend, kbeg, a_full, tid);
                    sb[s_new].load(g.B, g.ldb, n0, g.N, kbeg + (it + NSTG - 1) * BK, kend, kbeg, b_full, tid);
                }
                if (it + 1 < nt) {
                    wait_tiles(min(it + NSTG - 1, nt - 1) - (it + 1));
                    float* Ln = ((it + 1) & 1) ? L1 : L0;
                    sa[s_nxt].pin();
                    sb[s_nxt].pin();
                    sa[s_nxt].store(Ln, tid);
                    sb[s_nxt].store(Ln + GA::FLOATS, tid);
                }
                __syncthreads();
            }
        }
        GD_WAIT_VM(0);
        return;  // loader waves take no part in the epilogue (a finished wave leaves the barrier count)
    }
#undef GD_WAIT_VM

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef GD_STAMP  // tools/gemm_probe.hip diagnostic build only (-DGD_STAMP): s_memtime stamps of one wave of every 269th
                 // workgroup into g.rowpart -- per k-step: 0 after the barrier, 1 fragments in registers, 2 MFMAs issued,
                 // 3 barrier passed.  The forced lgkmcnt(0) makes this build ~15 % slower than the real kernel.
    unsigned long long* stamp_buf = reinterpret_cast<unsigned long long*>(g.rowpart);
    const int stamp_slot = (stamp_buf && (blockIdx.x % 269) == 0 && tid == 0) ? (int)(blockIdx.x / 269) : -1;
#define GD_ST(K) do { if (stamp_slot >= 0 && it < 32) stamp_buf[(stamp_slot * 32 + it) * 4 + (K)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GD_ST(K) ((void)0)
#endif
    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        const float* As = (it & 1) ? L1 : L0;
        const float* Bs = As + GA::FLOATS;
        GD_ST(0);
#pragma unroll
        for (int c = 0; c < BK / 16; ++c) {
            float fa[TM][4], fb[TN][4];
            load_frag<LAYA, BM, BK, TM>(As, wm0, c, r, q, fa);
            load_frag<LAYB, BN, BK, TN>(Bs, wn0, c, r, q, fb);
#ifdef GD_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GD_ST(1);
#endif
#pragma unroll
            for (int sg = 0; sg < 4; ++sg)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][sg], fb[j][sg], acc[i][j], 0, 0, 0);
        }
        GD_ST(2);
        __syncthreads();
        GD_ST(3);
    }
#undef GD_ST
    if constexpr (EPI == GD_EPI_ADAMW)  // plain stores measured faster without the LDS round trip (0.260 vs 0.280 ms)
        gemm_epilogue_rows<BM, BN, TM, TN, WAVES_M, WAVES_N, EPI, spec_lds_floats<LAYA, LAYB, BM, BN, BK, WAVES_M, EPI>(),
                           NTHREADS>(acc, g, m0, n0, wn0, r, q, wave, tid, smem);
    else
        gemm_epilogue<BM, TM, TN, WAVES_N, EPI>(acc, g, m0, n0, wm0, wn0, r, q, split, tile_n, wave, tid, smem);
}

template <int LAYA, int LAYB, int BM, int BN, int BK, int WM, int WN, int EPI>
int launch_one(GdGemm& g, hipStream_t s) {
    using GA = TileGeom<LAYA, BM, BK>;
    using GB = TileGeom<LAYB, BN, BK>;
    size_t lds = (size_t)2 * (GA::FLOATS + GB::FLOATS) * sizeof(float);
    // Row-contiguous x row-contiguous products (the weight gradients) run the wave-specialised kernel: measured
    // +6-7 % at 128x128 tiles on the Yelp shape, no gain or a loss for the K-contiguous products
    // (profiles/r01_spec_ab.txt).  GDMCF_GEMM_SPEC=0 switches it off for A/B runs.
    static const bool spec_on = !(getenv("GDMCF_GEMM_SPEC") && atoi(getenv("GDMCF_GEMM_SPEC")) == 0);
    constexpr bool SPEC_OK = (LAYA == 1 && LAYB == 1 && BM == 128 && BN == 128);
    const bool spec = SPEC_OK && spec_on;
    if (spec) lds = (size_t)spec_lds_floats<LAYA, LAYB, BM, BN, BK, WM, EPI>() * sizeof(float);
    void (*kern)(const GdGemm) = gemm_f32_kernel<LAYA, LAYB, BM, BN, BK, WM, WN, EPI>;
    if constexpr (SPEC_OK) {
        // register stages of the loader waves: 2 (default) or 4 via GDMCF_SPEC_STAGES; measured 0.260 / 0.262 / 0.257
        // / 0.272 ms for 2 / 3 / 4 / 6 stages on the Yelp dW products -- the kernel is not load-latency bound
        static const int nstg = getenv("GDMCF_SPEC_STAGES") ? atoi(getenv("GDMCF_SPEC_STAGES")) : 2;
        if (spec) {
            kern = gemm_f32_spec_kernel<LAYA, LAYB, BM, BN, BK, WM, WN, EPI, 2>;
            // (four stages of 32-deep tiles do not fit the loader waves' registers: with asm loads a spilled stage register
            // would be read before its data arrives -- the build refuses kernels of this file that spill, gdmcf_amd/build.py)
            if constexpr (BK == 16) {
                if (nstg == 4) kern = gemm_f32_spec_kernel<LAYA, LAYB, BM, BN, BK, WM, WN, EPI, 4>;
            }
        }
    }
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(spec ? 2 * NTHREADS : NTHREADS), lds, s, g);
    }
    return gd_launch_status("gemm_f32");
}

template <int LAYA, int LAYB, int BK, int EPI>
int launch_class(int cls, GdGemm& g, hipStream_t s) {
    switch (cls) {
        case 0: return launch_one<LAYA, LAYB, 80, 128, BK, 1, 4, EPI>(g, s);
        case 1: return launch_one<LAYA, LAYB, 128, 128, BK, 2, 2, EPI>(g, s);
        case 2: return launch_one<LAYA, LAYB, 64, 64, BK, 2, 2, EPI>(g, s);
    }
    gdmcf_set_error("bad gemm shape class %d", cls);
    return GDMCF_E_ARG;
}

}  // namespace

// (class 3: bf16 only; class 4 = 208x128: three-term split only -- gemm_split.hip)
int gd_gemm_tile_m(int cls) { return cls == 0 ? 80 : (cls == 1 ? 128 : (cls == 2 ? 64 : 208)); }
int gd_gemm_tile_n(int cls) { return cls == 2 ? 64 : (cls == 3 ? 256 : 128); }
int gd_gemm_bk(int layA, int layB) { return (layA == GD_LAY_MC && layB == GD_LAY_MC) ? 16 : 32; }

int gd_pick_shape_class(int M, int N) {
    if (M <= 64 || N <= 64) return 2;
    // 128-row tiles reuse each B fragment over more rows (measured 101 vs 92 TF on the dW products); the
    // 80-row tile exists for batch-sized M such as 400 = 5 x 80 where 128 would pad by 28 %.
    const long pad80 = (long)gd_cdiv(M, 80) * 80, pad128 = (long)gd_cdiv(M, 128) * 128;
    return (pad128 * 100 <= pad80 * 103) ? 1 : 0;
}

int gd_gemm_launch(int layA, int layB, int epi, int cls, GdGemm& g, hipStream_t s) {
    if (g.bf16 == 2) { t_gd_last_gemm = 8; return gd_gemm_split_launch(layA, layB, epi, cls, g, s); }
    if (g.bf16) { t_gd_last_gemm = 7; return gd_gemm_bf16_launch(layA, layB, epi, cls, g, s); }
    {
        const int rc = gd_gemm_dr_launch(layA, layB, epi, g, s);  // barrier-free register-streaming kernels where they apply
        if (rc != GD_DR_NOT_TAKEN) return rc;
    }
    t_gd_last_gemm = 1;
    // the branch-free edge loader clamps 16-byte vectors onto valid elements: it needs >= 4 elements along the
    // contiguous axis of every operand (K for K-contiguous operands, rows for row-contiguous ones); anything smaller
    // goes to the element-wise kernel of gemm_small.hip
    if (((layA == GD_LAY_KC || layB == GD_LAY_KC) && g.K < 4) || (layA == GD_LAY_MC && g.M < 4) ||
        (layB == GD_LAY_MC && g.N < 4))
        { t_gd_last_gemm = 9; return gd_gemm_small_launch(layA, layB, epi, g, s); }
    if (layA == GD_LAY_KC && layB == GD_LAY_KC) {
        switch (epi) {
            case GD_EPI_SLAB: return launch_class<GD_LAY_KC, GD_LAY_KC, 32, GD_EPI_SLAB>(cls, g, s);
            case GD_EPI_BIAS_ACT: return launch_class<GD_LAY_KC, GD_LAY_KC, 32, GD_EPI_BIAS_ACT>(cls, g, s);
            case GD_EPI_LOSS: return launch_class<GD_LAY_KC, GD_LAY_KC, 32, GD_EPI_LOSS>(cls, g, s);
            case GD_EPI_POST: return launch_class<GD_LAY_KC, GD_LAY_KC, 32, GD_EPI_POST>(cls, g, s);
        }
    } else if (layA == GD_LAY_KC && layB == GD_LAY_MC) {
        if (epi == GD_EPI_SLAB) return launch_class<GD_LAY_KC, GD_LAY_MC, 32, GD_EPI_SLAB>(cls, g, s);
    } else if (layA == GD_LAY_MC && layB == GD_LAY_MC) {
        static const int tn_bk = getenv("GDMCF_TN_BK") ? atoi(getenv("GDMCF_TN_BK")) : 16;  // tuning knob
        if (epi == GD_EPI_STORE && tn_bk == 32) return launch_class<GD_LAY_MC, GD_LAY_MC, 32, GD_EPI_STORE>(cls, g, s);
        if (epi == GD_EPI_STORE) return launch_class<GD_LAY_MC, GD_LAY_MC, 16, GD_EPI_STORE>(cls, g, s);
        if (epi == GD_EPI_ADAMW) return launch_class<GD_LAY_MC, GD_LAY_MC, 16, GD_EPI_ADAMW>(cls, g, s);
    }
    gdmcf_set_error("unsupported gemm variant (layA=%d layB=%d epi=%d)", layA, layB, epi);
    return GDMCF_E_UNSUPPORTED;
}
