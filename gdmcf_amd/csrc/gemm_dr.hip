// "Direct-to-register" f32 MFMA products: no LDS, no barriers, independent persistent waves.
//
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 matrix rate, so an operand byte is worth sixteen times more matrix
// time than in a bf16 kernel: a wave can afford to fetch its OWN operands from L1/L2 straight into the MFMA register
// layout.  What that buys (measured, tools/dr_probe.hip): no LDS round trip, no workgroup barrier -- hence
// no lockstep between waves, a wave that waits for memory or stores its results leaves the matrix pipe to its SIMD
// partner -- output tiles small enough (64 x 64 per wave) to balance 8 600 of them over 1 024 SIMDs from a ticket counter,
// and results that leave the accumulators as 16 contiguous bytes per lane.
//
// Row-contiguous operand P[k][rows] (the weight-gradient products, reference main.py:350 / models/DNN.py:79-86): one
// buffer_load_dwordx4 per wave brings rows r0 .. r0+63 of four consecutive k; lane (i = lane & 15, q = lane >> 4) holds
// P[k0 + q][r0 + 4 i + e], e = 0..3, and register e IS the operand of the MFMA block whose 16 rows are r0 + 4 i + e (any
// fixed assignment of matrix rows to MFMA rows is as good as another).  256 contiguous bytes per lane group, nothing to
// transpose.  With both operands loaded this way the accumulators hold
//     acc[a][e][b][f][t] = C[m0 + 64 a + 16 q + 4 t + e][n0 + 64 b + 4 r + f],   r = lane & 15, q = lane >> 4,
// i.e. four consecutive columns (f) per lane and 256 contiguous bytes per row and store instruction.
//
// Pipeline (per wave): a ring of R = D + 1 register slots, one k-step (4 k) each.  While step s is multiplied, the loads of
// step s + D are issued BETWEEN its MFMAs into the slot step s - 1 has just left; every step waits with the same counted
// s_waitcnt vmcnt.  Loads go through raw buffer descriptors (base + per-lane voffset + scalar soffset): advancing a k-step is
// one scalar add, and anything outside the matrix returns 0 instead of faulting (K tails, steps past the end).  The ring runs
// CONTINUOUSLY across tiles -- during the last D steps of a tile the loads already belong to the next tile, whose id comes
// from a ticket drawn one tile earlier -- so a tile boundary costs neither a pipeline fill nor a drain; a tile runs a
// multiple of R steps so that slot indices stay compile-time constants.  Two waves share a SIMD (one 512-thread workgroup
// per CU); the hardware prefers the older one, which starves the younger and leaves a long tail, so a wave raises its
// priority with the progress of its tile (s_setprio): the tile closest to its end wins.
//
// Determinism: the tile -> wave assignment is dynamic, the arithmetic of a tile is not (fixed k order, one wave per tile):
// results are bit-identical from run to run.
#include <stdlib.h>

#include "common.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned int g_dr_ticket[32];  // one counter per call site (GdGemm::prof_tag); zero between launches

__device__ __forceinline__ i32x4 dr_srd(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

// ring loads are asm: hipcc neither counts nor waits for them, the kernel places the counted s_waitcnt itself
__device__ __forceinline__ f32x4 dr_load(i32x4 srd, uint32_t voff, uint32_t soff) {
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(srd), "s"(soff) : "memory");
    return v;
}

template <int N>
__device__ __forceinline__ void dr_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// lane 0 draws a ticket.  The returning atomic is asm so that hipcc does not wait for it on the spot: it is older than
// every ring load of the tile it is issued in and has long landed when the cursor leaves that tile.
__device__ __forceinline__ unsigned int dr_ticket_issue(unsigned int* ctr) {
    unsigned int t;
    unsigned long long save;
    const unsigned int zero = 0, one = 1;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                 : "=&v"(t), "=&s"(save) : "v"(zero), "v"(one), "s"(ctr) : "memory");
    return t;
}

struct DrArgs {
    GdGemm g;
    int tiles_m, tiles_n, m_fastest;
    int ksp;      // k-steps run per tile (a multiple of the ring size; steps past K load zeros)
    int ctr;      // index into g_dr_ticket
    int stagger;  // waves 4-7 of a workgroup start this many x 3.4 us later
};

// C[M,N] = A[K,M]^T * B[K,N], both operands row-contiguous.  TA / TB: 64-row load units per operand and k-step.
template <int TA, int TB, int D, int EPI>
__global__ __launch_bounds__(512, 2) void dr_tn_kernel(const DrArgs d) {
    static_assert(EPI == GD_EPI_STORE || EPI == GD_EPI_ADAMW, "weight-gradient products");
    constexpr int LPS = TA + TB;  // loads per k-step
    constexpr int R = D + 1;
    static_assert(LPS * D <= 63, "vmcnt is a 6-bit counter");
    const GdGemm& g = d.g;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int ntiles = d.tiles_m * d.tiles_n;
    unsigned int* ctr = &g_dr_ticket[d.ctr];
    const int n_waves = gridDim.x * 8;
    int cur = __builtin_amdgcn_readfirstlane(blockIdx.x * 8 + (threadIdx.x >> 6));  // first tile: static; then n_waves + ticket
    if (cur >= ntiles) return;
    if (d.stagger > 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256)
        for (int i = 0; i < d.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    const int KSP = d.ksp;
    // the descriptors end with the last valid element: rows k >= K and everything behind the matrices reads as 0
    const i32x4 srdA = dr_srd(g.A, (uint32_t)(((int64_t)(g.K - 1) * g.lda + g.M) * 4));
    const i32x4 srdB = dr_srd(g.B, (uint32_t)(((int64_t)(g.K - 1) * g.ldb + g.N) * 4));
    const uint32_t sa = 16u * (uint32_t)g.lda, sb = 16u * (uint32_t)g.ldb;
    const uint32_t c_bytes = (uint32_t)(((int64_t)(g.M - 1) * g.ldc + g.N) * 4);
    const __amdgpu_buffer_rsrc_t srdC = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)c_bytes, 0x00020000);

    // ---- load cursor: the tile whose operands are being fetched ----
    uint32_t offA[TA], offB[TB];  // per-lane byte offsets inside the cursor's tile
    uint32_t ka = 0, kb = 0;      // byte offset of the k-step to load next (soffset operand)
    int l_left = KSP;             // steps of the cursor's tile not yet issued
    auto set_cursor = [&](int tile) {
        const bool ok = tile < ntiles;  // past the end the cursor is parked outside both matrices: every load returns 0
        const int tm = d.m_fastest ? (tile % d.tiles_m) : (tile / d.tiles_n);
        const int tn = d.m_fastest ? (tile / d.tiles_m) : (tile % d.tiles_n);
#pragma unroll
        for (int a = 0; a < TA; ++a) offA[a] = ok ? (uint32_t)(q * g.lda + tm * (64 * TA) + 64 * a + 4 * r) * 4u : 0xFFFFFFF0u;
#pragma unroll
        for (int b = 0; b < TB; ++b) offB[b] = ok ? (uint32_t)(q * g.ldb + tn * (64 * TB) + 64 * b + 4 * r) * 4u : 0xFFFFFFF0u;
        ka = kb = 0;
        l_left = KSP;
    };
    set_cursor(cur);
    f32x4 ra[R][TA], rb[R][TB];
#pragma unroll
    for (int u = 0; u < D; ++u) {
#pragma unroll
        for (int a = 0; a < TA; ++a) ra[u][a] = dr_load(srdA, offA[a], ka);
#pragma unroll
        for (int b = 0; b < TB; ++b) rb[u][b] = dr_load(srdB, offB[b], kb);
        ka += sa;
        kb += sb;
        --l_left;
    }
    const int q1 = (KSP / R / 4) * R, q2 = (KSP / R / 2) * R, q3 = (KSP / R * 3 / 4) * R;
    for (;;) {
        unsigned int tick = dr_ticket_issue(ctr);  // id of the tile AFTER this one: needed when the cursor leaves this tile
        int nxt = 0;
        const int tm = d.m_fastest ? (cur % d.tiles_m) : (cur / d.tiles_n);
        const int tn = d.m_fastest ? (cur / d.tiles_m) : (cur % d.tiles_n);
        const int m0 = tm * 64 * TA, n0 = tn * 64 * TB;
        f32x4 acc[TA][4][TB][4];
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < TB; ++b)
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[a][e][b][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(0);
        for (int s0 = 0; s0 < KSP; s0 += R) {
            if (s0 == q1) __builtin_amdgcn_s_setprio(1);
            else if (s0 == q2) __builtin_amdgcn_s_setprio(2);
            else if (s0 == q3) __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int u = 0; u < R; ++u) {
                constexpr int NM = 16 * TA * TB;  // MFMAs of this step; the LPS loads ride behind MFMA 2, 6, 10, ...
                const int v = (u + D) % R;        // slot of step s + D (= the slot step s - 1 has left)
                dr_wait<LPS*(D - 1)>();           // step s has landed; steps s+1 .. s+D-1 stay in flight
#pragma unroll
                for (int a = 0; a < TA; ++a) asm volatile("" : "+v"(ra[u][a]));
#pragma unroll
                for (int b = 0; b < TB; ++b) asm volatile("" : "+v"(rb[u][b]));
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    const int a = i / (16 * TB), e = (i / (4 * TB)) % 4, b = (i / 4) % TB, f = i % 4;
                    acc[a][e][b][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u][a][e], rb[u][b][f], acc[a][e][b][f], 0, 0, 0);
                    if (i % 4 == 1 && i / 4 < LPS) {
                        __builtin_amdgcn_sched_barrier(0);
                        const int l = i / 4;
                        if (l < TA) ra[v][l] = dr_load(srdA, offA[l], ka);
                        else rb[v][l - TA] = dr_load(srdB, offB[l - TA], kb);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (i == 4 * LPS + 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        ka += sa;
                        kb += sb;
                        if (--l_left == 0) {  // once per tile: the cursor moves on to the next tile
                            // the ticket was issued at step 0 of this tile: it is older than every load the counted waits leave
                            // in flight once D - 1 later steps have issued theirs, i.e. when the tile runs >= 2 D steps
                            if (KSP < 2 * D + 2) dr_wait<0>();
                            asm volatile("" : "+v"(tick));
                            const int tk = __builtin_amdgcn_readfirstlane(tick);
                            // exactly `ntiles` tickets are drawn per launch (one per processed tile): the last one resets
                            if (tk == ntiles - 1 && lane == 0) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            nxt = n_waves + tk;
                            set_cursor(nxt);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: 16 bytes per lane along N; the row part of the address is scalar (soffset); rows past M fall outside
        // the descriptor, columns past N are cut by the lane ----
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int n = n0 + 64 * b + 4 * r;
            const uint32_t vo = (uint32_t)(16 * q * g.ldc + n) * 4u;
            if (n + 3 < g.N) {
#pragma unroll
                for (int a = 0; a < TA; ++a)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            f32x4 gv = {acc[a][e][b][0][t], acc[a][e][b][1][t], acc[a][e][b][2][t], acc[a][e][b][3][t]};
                            const uint32_t so = (uint32_t)(m0 + 64 * a + 4 * t + e) * (uint32_t)g.ldc * 4u;
                            if (EPI == GD_EPI_STORE) {
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gv), srdC, vo, so, 0);
                            } else {
                                const __amdgpu_buffer_rsrc_t srdM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.aux), 0, (int)c_bytes, 0x00020000);
                                const __amdgpu_buffer_rsrc_t srdV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.aux2), 0, (int)c_bytes, 0x00020000);
                                f32x4 pv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdC, vo, so, 0));
                                f32x4 mv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdM, vo, so, 0));
                                f32x4 vv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdV, vo, so, 0));
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    float pk = pv[k], mk = mv[k], vk = vv[k];
                                    gd_adam_elem(pk, gv[k], mk, vk, g.adam);
                                    pv[k] = pk;
                                    mv[k] = mk;
                                    vv[k] = vk;
                                }
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pv), srdC, vo, so, 0);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mv), srdM, vo, so, 0);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vv), srdV, vo, so, 0);
                            }
                        }
            } else if (n < g.N) {  // the lane's four columns straddle N (last column tile only)
                float* __restrict__ Mo = const_cast<float*>(g.aux);
                float* __restrict__ Vo = const_cast<float*>(g.aux2);
                for (int a = 0; a < TA; ++a)
                    for (int t = 0; t < 4; ++t)
                        for (int e = 0; e < 4; ++e) {
                            const int m = m0 + 64 * a + 16 * q + 4 * t + e;
                            if (m >= g.M) continue;
                            for (int k = 0; k < 4; ++k) {
                                if (n + k >= g.N) continue;
                                const int64_t o = (int64_t)m * g.ldc + n + k;
                                const float gk = acc[a][e][b][k][t];
                                if (EPI == GD_EPI_STORE) {
                                    g.C[o] = gk;
                                } else {
                                    float pk = g.C[o], mk = Mo[o], vk = Vo[o];
                                    gd_adam_elem(pk, gk, mk, vk, g.adam);
                                    g.C[o] = pk;
                                    Mo[o] = mk;
                                    Vo[o] = vk;
                                }
                            }
                        }
            }
        }
        if (nxt >= ntiles) break;
        cur = nxt;
    }
    // the parked cursor's loads are still in flight: their destination registers stay live (and untouched by the compiler)
    // until they have landed -- a register hipcc believes dead and reuses would be overwritten by such a load
    dr_wait<0>();
#pragma unroll
    for (int u = 0; u < R; ++u) {
#pragma unroll
        for (int a = 0; a < TA; ++a) asm volatile("" ::"v"(ra[u][a]));
#pragma unroll
        for (int b = 0; b < TB; ++b) asm volatile("" ::"v"(rb[u][b]));
    }
}

template <int D, int EPI>
void dr_tn_go(const DrArgs& d, hipStream_t s) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    hipLaunchKernelGGL((dr_tn_kernel<1, 1, D, EPI>), dim3(n_cu), dim3(512), 0, s, d);
}

}  // namespace

// Returns GD_DR_NOT_TAKEN when the product is not one this file handles (the caller falls back to the LDS-tiled kernels).
int gd_gemm_dr_launch(int layA, int layB, int epi, GdGemm& g, hipStream_t s) {
    static const bool on = !(getenv("GDMCF_GEMM_DR") && atoi(getenv("GDMCF_GEMM_DR")) == 0);
    if (!on || g.bf16) return GD_DR_NOT_TAKEN;
    if (!(layA == GD_LAY_MC && layB == GD_LAY_MC && (epi == GD_EPI_STORE || epi == GD_EPI_ADAMW))) return GD_DR_NOT_TAKEN;
    if (g.accumulate || g.splits > 1 || g.C16) return GD_DR_NOT_TAKEN;
    // 32-bit byte offsets inside every matrix; enough tiles for the persistent waves to be worth it
    const int64_t lim = (int64_t)1 << 32;
    if ((int64_t)g.K * g.lda * 4 >= lim || (int64_t)g.K * g.ldb * 4 >= lim || (int64_t)g.M * g.ldc * 4 >= lim) return GD_DR_NOT_TAKEN;
    if (g.lda < g.M || g.ldb < g.N || g.ldc < g.N || g.K < 1) return GD_DR_NOT_TAKEN;
    const long tiles = (long)gd_cdiv(g.M, 64) * gd_cdiv(g.N, 64);
    if (tiles < 512 || g.K < 128) return GD_DR_NOT_TAKEN;  // (short reductions: a tile is all prologue; the LDS-tiled kernels take them)
    DrArgs d = {};
    d.g = g;
    d.tiles_m = gd_cdiv(g.M, 64);
    d.tiles_n = gd_cdiv(g.N, 64);
    d.m_fastest = d.tiles_m <= d.tiles_n;  // tiles that share the LARGER operand's panel draw consecutive tickets
    d.ctr = g.prof_tag & 31;
    static const int stagger = getenv("GDMCF_DR_STAGGER") ? atoi(getenv("GDMCF_DR_STAGGER")) : 3;
    d.stagger = stagger;
    const int ks = gd_cdiv(g.K, 4);
    // ring depth: the one whose size wastes the fewest padded steps per tile
    int best = 9, waste = 1 << 30;
    for (int dd : {9, 8, 7}) {
        const int w = gd_cdiv(ks, dd + 1) * (dd + 1) - ks;
        if (w < waste) { waste = w; best = dd; }
    }
    d.ksp = ks + waste;
    g.tiles_m = d.tiles_m;
    g.tiles_n = d.tiles_n;
    {
        GdProfScope prof(g.prof_tag, 2.0 * g.M * g.N * g.K, s);
#define GD_DR_GO(DD)                                                      \
    do {                                                                  \
        if (epi == GD_EPI_STORE) dr_tn_go<DD, GD_EPI_STORE>(d, s);        \
        else dr_tn_go<DD, GD_EPI_ADAMW>(d, s);                            \
    } while (0)
        if (best == 9) GD_DR_GO(9);
        else if (best == 8) GD_DR_GO(8);
        else GD_DR_GO(7);
#undef GD_DR_GO
    }
    return gd_launch_status("gemm_dr");
}
