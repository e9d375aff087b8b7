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
typedef f32x4 f32x4_dr_u __attribute__((aligned(4)));

// ticket counters: one SET per launch in flight (dr_ticket_slot below), one queue per XCD inside a set, each on a 128-byte line of
// its own; a queue's last draw resets it, so every set is zero again when its launch has drained
__device__ unsigned int g_dr_ticket[32][8][32];

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

#ifndef GD_ADAMW_DBG
#define GD_ADAMW_DBG 0
#endif
#ifndef DR_ADAM_UB
#define DR_ADAM_UB 2  // row groups per batch of the fused-AdamW epilogue pipeline (dr_tn_kernel)
#endif

struct DrArgs {
    GdGemm g;
    int tiles_m, tiles_n, m_fastest;
    int ksp;      // k-steps run per tile (a multiple of the ring size; steps past K load zeros)
    int ctr;      // index into g_dr_ticket (dr_ticket_slot: a set of its own for every launch that may be in flight)
    const GdAdamHyper* adam_dev;  // fused AdamW: this step's scalars in device memory (graph replay), NULL = GdGemm::adam
    int stagger;  // waves 4-7 of a workgroup start this many x 3.4 us later
};

// C[M,N] = A[K,M]^T * B[K,N], both operands row-contiguous.  TA / TB: 64-row load units per operand and k-step.
template <int TA, int TB, int D, int EPI>
__global__ __launch_bounds__(512, 2) void dr_tn_kernel(const DrArgs d) {
    static_assert(EPI == GD_EPI_STORE, "plain weight-gradient product (the fused-AdamW form: dr_tn_adamw_kernel)");
    constexpr int LPS = TA + TB;  // loads per k-step
    constexpr int R = D + 1;
    static_assert(LPS * D <= 63, "vmcnt is a 6-bit counter");
    const GdGemm& g = d.g;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    // ---- tiles and tickets.  "Panel" = the tiles that share a 64-row slice of the LARGER operand; the panels p with p % 8 == x
    // form queue x, served first by the waves that run on XCD x (HW_REG_XCC_ID): a panel's slice is then fetched into ONE L2
    // instead of eight (measured before: 325 MB fetched per launch for 57 MB of operands).  A wave that finds its queue empty
    // goes on to the next one, so the queues only set who takes what first, never who may take what (placement-independent).
    // Ticket t of queue x = tile (t % minor) of its panel (t / minor).  Every wave draws from a queue until a draw fails, i.e.
    // fails exactly once per queue: a queue of n tiles sees n + n_waves draws per launch and the last draw resets it.
    const int n_waves = gridDim.x * 8;
    const int minor = d.m_fastest ? d.tiles_m : d.tiles_n;   // tiles per panel
    const int panels = d.m_fastest ? d.tiles_n : d.tiles_m;
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    int qx = xcc & 7;                                         // queue being drawn from
    int visited = 0;                                          // queues this wave has exhausted
    auto q_tiles = [&](int x) { return ((panels - x + 7) >> 3) * minor; };
    auto tile_of = [&](int x, int t) {                        // -> tile id in the (m_fastest ? tn * tiles_m + tm : tm * tiles_n + tn) numbering
        const int p = (t / minor) * 8 + x, i = t % minor;
        return p * minor + i;
    };
    // draw (blocking) until a queue yields a tile or all eight have failed; returns -1 when the wave is done
    auto draw_blocking = [&]() {
        for (;;) {
            if (visited == 8) return -1;
            unsigned int* c = &g_dr_ticket[d.ctr][qx][0];
            unsigned int tk = dr_ticket_issue(c);
            dr_wait<0>();
            asm volatile("" : "+v"(tk));
            const int t = __builtin_amdgcn_readfirstlane(tk);
            const int n = q_tiles(qx);
            if (t == n + n_waves - 1 && lane == 0) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < n) return tile_of(qx, t);
            qx = (qx + 1) & 7;
            ++visited;
        }
    };
    if (d.stagger > 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256)
        for (int i = 0; i < d.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    int cur = draw_blocking();
    if (cur < 0) return;
    const int ntiles = d.tiles_m * d.tiles_n;
    const int KSP = d.ksp;
    // the descriptors end with the last valid element: rows k >= K and everything behind the matrices reads as 0
    const i32x4 srdA = dr_srd(g.A, (uint32_t)(((int64_t)(g.K - 1) * g.lda + g.M) * 4));
    const i32x4 srdB = dr_srd(g.B, (uint32_t)(((int64_t)(g.K - 1) * g.ldb + g.N) * 4));
    const uint32_t sa = 16u * (uint32_t)g.lda, sb = 16u * (uint32_t)g.ldb;
    // (with a bias column the product has one more column than C: the descriptor ends with C's own last element, or the first
    // lane of the row tile past M would pass the range check by that one element)
    const uint32_t c_bytes = (uint32_t)(((int64_t)(g.M - 1) * g.ldc + (g.out2 ? g.N - 1 : g.N)) * 4);
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
        // the ticket of the tile AFTER this one travels under this tile's work (needed when the cursor leaves this tile)
        unsigned int* tctr = &g_dr_ticket[d.ctr][qx][0];
        unsigned int tick = visited < 8 ? dr_ticket_issue(tctr) : 0u;
        const bool drew = visited < 8;
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
                            nxt = -1;
                            if (drew) {
                                if (KSP < 2 * D + 2) dr_wait<0>();
                                asm volatile("" : "+v"(tick));
                                const int tk = __builtin_amdgcn_readfirstlane(tick);
                                const int nq = q_tiles(qx);
                                if (tk == nq + n_waves - 1 && lane == 0) __hip_atomic_store(tctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (tk < nq) {
                                    nxt = tile_of(qx, tk);
                                } else {  // this queue is empty (a few times per wave, at the end of the launch): try the others
                                    qx = (qx + 1) & 7;
                                    ++visited;
                                    nxt = draw_blocking();
                                }
                            }
                            if (nxt < 0) nxt = ntiles;  // parks the cursor
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
            if (n + 3 < (g.out2 ? g.N - 1 : g.N)) {  // (a bias column, the last one, never goes out with a 16-byte group)
                // The row offset travels in the VGPR offset, the scalar offset field stays 0.  With an SGPR there hipcc emits
                // `buffer_store_dwordx4 v[146:149], v0, s[36:39], s10 offen` and refills v146..149 for the next row in the very
                // next instruction: LLVM's hazard recognizer holds that a store of more than 64 bits needs no wait state before
                // its data registers are rewritten when soffset is a register -- on gfx950 it does: lanes 12-15 of every
                // 16-lane row of the FIRST data register went out with the next row's values, in timing-dependent launches
                // (DESIGN 4.1b).  With soffset = 0 the recognizer inserts the s_nop itself.
#pragma unroll
                for (int a = 0; a < TA; ++a)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            f32x4 gv = {acc[a][e][b][0][t], acc[a][e][b][1][t], acc[a][e][b][2][t], acc[a][e][b][3][t]};
                            const uint32_t so = (uint32_t)(m0 + 64 * a + 4 * t + e) * (uint32_t)g.ldc * 4u;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gv), srdC, vo + so, 0, 0);
                        }
            } else if (n < g.N) {  // the lane's four columns straddle N (last column tile only)
                for (int a = 0; a < TA; ++a)
                    for (int t = 0; t < 4; ++t)
                        for (int e = 0; e < 4; ++e) {
                            const int m = m0 + 64 * a + 16 * q + 4 * t + e;
                            if (m >= g.M) continue;
                            for (int k = 0; k < 4; ++k) {
                                if (n + k >= g.N) continue;
                                const float gk = acc[a][e][b][k][t];
                                if (g.out2 && n + k == g.N - 1) g.out2[m] = gk;  // the bias column (operand B's extra column): its own vector
                                else g.C[(int64_t)m * g.ldc + n + k] = gk;
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

// ---------------------------------------------------------------------------------------------------------------------
// The weight-gradient product with the AdamW update of that weight in the SAME kernel (reference main.py:350-351:
// loss.backward(); optimizer.step()), single GPU: G = A^T B never reaches memory, W / exp_avg / exp_avg_sq are read and written
// once -- 24 B per parameter instead of 32 + the separate pass.
//
// Round 3 ran the update as the tile's epilogue.  Measured in round 4 (tools/fused_probe.py): the optimiser stream of the
// launch (826 MB at the Yelp shape) then runs at the full HBM rate -- and the matrix pipe stands still meanwhile: fused time =
// matrix time + stream time for every reduction length (0.089 + 0.129 ms at K = 128, 0.215 + 0.110 ms at K = 400), i.e. no
// overlap at all, with one memory round trip per row group or with several in flight alike.  All waves reach their
// epilogues in phase (same tile length everywhere), the burst saturates HBM, and the waves that should multiply meanwhile wait
// for operands behind it (vmcnt retires in order; the per-CU memory pipeline queues their L2 hits behind the misses).
//
// Here the stream of tile i runs INSIDE the k loop of tile i + 1 of the same wave, at a fixed pace: the finished tile is parked
// in LDS (16 KB per wave, wave-private: no barrier), and every ring round (R k-steps) updates two of its sixteen row groups
// (4 rows x 256 B of each of the three arrays): the three loads of a group are issued between the MFMAs of one k-step and
// consumed four steps later -- gradient from LDS, update, three stores.  The traffic is spread evenly over the matrix time of
// every wave (3.8 TB/s for the Yelp weights, 60 % of what HBM sustains), no bursts, whatever the phases of the waves.
//
// Everything vector-memory in the loop is inline asm with hand-counted waits (as the operand ring): hipcc's own counted waits see
// only its own instructions (guide 5.7).  vmcnt retires in order, so the extra instructions only shift the counts: the top-of-step
// wait allows 2 (D - 1) ring loads + the 12 optimiser instructions of a round minus those of the step itself, and the stream's
// instructions are issued ALWAYS -- parked outside the descriptor (loads return 0 without a fetch, stores are dropped) while no
// tile is pending -- so that the counts are the same in every round; the ring fill issues the parked instructions a previous
// round would have issued.  build.py:lint_vmcnt verifies every count and that nothing touches a register in flight.
// (A wave-specialised form -- eight multiplying waves + four stream waves per CU, hand-over through LDS, so that the stream's HBM
// accesses do not sit in the multiplying waves' in-order vmcnt -- was built and measured in round 4: bit-identical, and 0.02-0.03 ms
// SLOWER per launch; with its stream parked it costs only +0.015 ms over the plain product, so what the real stream costs is memory-
// system contention, not the counter.  profiles/r04_fused_stream_ablations.txt, section G.)
// Tiles whose lanes do not all own a full 16-byte group (the last column panel when N % 64 != 0, or with the bias column) are
// updated on the spot from the accumulators, as in round 3 (1/16 of the Yelp output-layer tiles, 1/538 of the first layer's).
// ---------------------------------------------------------------------------------------------------------------------
#ifdef GD_NO_SNOP  // (probe builds only: the build's lint rejects the kernel without the guard)
#define GD_SNOP ""
#else
#define GD_SNOP "s_nop 4\n\t"
#endif
#ifndef GD_ADAMW_ST
#define GD_ADAMW_ST 1  // cache policy of the optimiser stream's stores: 1 = nt (0 default, 2 sc1, 3 sc0 sc1: probe builds)
#endif
__device__ __forceinline__ void dr_store(f32x4 v, i32x4 srd, uint32_t voff) {
    // (trailing s_nop: a store of more than 64 bits reads its data registers after issue; hipcc cannot see that this is a store.
    // Leading s_nop 4: with ~100 live scalars hipcc keeps descriptors in VGPR lanes and restores them with v_readlane right in
    // front of the statement; an SGPR written by a VALU instruction must not be read by a vector-memory one for 5 wait states)
#if GD_ADAMW_ST == 1
    asm volatile(GD_SNOP "buffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd) : "memory");
#elif GD_ADAMW_ST == 2
    asm volatile(GD_SNOP "buffer_store_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd) : "memory");
#elif GD_ADAMW_ST == 3
    asm volatile(GD_SNOP "buffer_store_dwordx4 %0, %1, %2, 0 offen sc0 sc1\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd) : "memory");
#else
    asm volatile(GD_SNOP "buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd) : "memory");
#endif
}
// (read-write operand: the destination stays ONE virtual register for the whole kernel, so hipcc has no new value to place at
// every load and no PHI copies to insert at loop back edges -- copies that would move a register whose load is in flight)
// Cache policy of the stream's loads.  NT (non-temporal: the line is not kept in L2) when every 256-byte piece of a row covers
// whole 128-byte lines -- rows of W / exp_avg / exp_avg_sq on 128-byte lines, FusedAdamW.fuse_into_backward seats them so --:
// nothing of a line is left for a neighbouring tile, and the stream stops evicting the operand panels (0.277 -> 0.263 and
// 0.301 -> 0.276 ms for the two Yelp products).  With rows that start anywhere the neighbouring tiles' pieces share their first and
// last line, and a line that is not kept is fetched from HBM twice: 0.32 -> 0.40 ms (profiles/r04_fused_stream_ablations.txt D, I).
#ifndef GD_ADAMW_LD
#define GD_ADAMW_LD 0  // probe builds: policy of the loads when NTL is false (0 default, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt)
#endif
template <bool NTL>
__device__ __forceinline__ void dr_load0_rw(f32x4& v, i32x4 srd, uint32_t voff) {
    if constexpr (NTL) {
        asm volatile(GD_SNOP "buffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "+v"(v) : "v"(voff), "s"(srd) : "memory");
        return;
    }
#if GD_ADAMW_LD == 1
    asm volatile(GD_SNOP "buffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "+v"(v) : "v"(voff), "s"(srd) : "memory");
#elif GD_ADAMW_LD == 2
    asm volatile(GD_SNOP "buffer_load_dwordx4 %0, %1, %2, 0 offen sc1" : "+v"(v) : "v"(voff), "s"(srd) : "memory");
#elif GD_ADAMW_LD == 3
    asm volatile(GD_SNOP "buffer_load_dwordx4 %0, %1, %2, 0 offen sc0 sc1" : "+v"(v) : "v"(voff), "s"(srd) : "memory");
#elif GD_ADAMW_LD == 4
    asm volatile(GD_SNOP "buffer_load_dwordx4 %0, %1, %2, 0 offen sc0 sc1 nt" : "+v"(v) : "v"(voff), "s"(srd) : "memory");
#else
    asm volatile(GD_SNOP "buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(v) : "v"(voff), "s"(srd) : "memory");
#endif
}

template <int D, bool NTL>
__global__ __launch_bounds__(512, 2) void dr_tn_adamw_kernel(const DrArgs d) {
    constexpr int LPS = 2;  // ring loads per k-step
    constexpr int R = D + 1;
    // the optimiser stream's schedule inside a ring round: slot A is consumed AND reloaded in step UA, slot B in step UB -- a row
    // group's loads have a whole round (R k-steps, ~4 us with two waves per SIMD) to land
    constexpr int UA = 1, UB = 1 + R / 2;
    constexpr int XS = 6;   // optimiser instructions of such a step: 3 stores + 3 loads
    constexpr int XR = 12;  // ... of a round
    static_assert(UB < R && UB > UA, "two distinct steps of a round");
    static_assert(LPS * R + XS <= 63, "vmcnt is a 6-bit counter");
    constexpr uint32_t PARK = 0xFFFFFF00u;  // outside every descriptor
    const GdGemm& g = d.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) float dr_lds[];
    f32x4* const stash = reinterpret_cast<f32x4*>(dr_lds + wave * 4096) + lane;  // [16 accumulators][64 lanes] x 16 B
    // this step's AdamW scalars: by value, or -- a step replayed from a hipGraph -- from the device's step state
    GdAdamHyper hy = g.adam;
    if (d.adam_dev) hy = *d.adam_dev;
    // ---- tiles and tickets: as dr_tn_kernel ----
    const int n_waves = gridDim.x * 8;
    const int minor = d.m_fastest ? d.tiles_m : d.tiles_n;
    const int panels = d.m_fastest ? d.tiles_n : d.tiles_m;
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    int qx = xcc & 7, visited = 0;
    // (panels dealt round-robin over the queues.  Contiguous blocks of panels per queue -- neighbouring panels on one XCD, close in
    // time, for the 128-byte lines two neighbouring tiles share -- measured slower: 0.314 / 0.327 against 0.293 / 0.318 ms.)
    auto q_tiles = [&](int x) { return ((panels - x + 7) >> 3) * minor; };
    auto tile_of = [&](int x, int t) { return ((t / minor) * 8 + x) * minor + t % minor; };
    auto draw_blocking = [&]() {
        for (;;) {
            if (visited == 8) return -1;
            unsigned int* c = &g_dr_ticket[d.ctr][qx][0];
            unsigned int tk = dr_ticket_issue(c);
            dr_wait<0>();
            asm volatile("" : "+v"(tk));
            const int t = __builtin_amdgcn_readfirstlane(tk);
            const int n = q_tiles(qx);
            if (t == n + n_waves - 1 && lane == 0) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < n) return tile_of(qx, t);
            qx = (qx + 1) & 7;
            ++visited;
        }
    };
    if (d.stagger > 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256)
        for (int i = 0; i < d.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    int cur = draw_blocking();
    if (cur < 0) return;
    const int ntiles = d.tiles_m * d.tiles_n;
    const int KSP = d.ksp;
    const i32x4 srdA = dr_srd(g.A, (uint32_t)(((int64_t)(g.K - 1) * g.lda + g.M) * 4));
    const i32x4 srdB = dr_srd(g.B, (uint32_t)(((int64_t)(g.K - 1) * g.ldb + g.N) * 4));
    const uint32_t sa = 16u * (uint32_t)g.lda, sb = 16u * (uint32_t)g.ldb;
    const int n_lim = g.out2 ? g.N - 1 : g.N;  // columns of C (a bias column, the last one of the product, goes to out2)
    const uint32_t c_bytes = (uint32_t)(((int64_t)(g.M - 1) * g.ldc + n_lim) * 4);
    const i32x4 srdW = dr_srd(g.C, c_bytes), srdM = dr_srd(g.aux, c_bytes), srdV = dr_srd(g.aux2, c_bytes);
    const uint32_t ldc4 = (uint32_t)g.ldc * 4u;

    uint32_t offA, offB, ka = 0, kb = 0;
    int l_left = KSP;
    auto set_cursor = [&](int tile) {
        const bool ok = tile < ntiles;
        const int tm = d.m_fastest ? (tile % d.tiles_m) : (tile / d.tiles_n);
        const int tn = d.m_fastest ? (tile / d.tiles_m) : (tile % d.tiles_n);
        offA = ok ? (uint32_t)(q * g.lda + tm * 64 + 4 * r) * 4u : 0xFFFFFFF0u;
        offB = ok ? (uint32_t)(q * g.ldb + tn * 64 + 4 * r) * 4u : 0xFFFFFFF0u;
        ka = kb = 0;
        l_left = KSP;
    };
    set_cursor(cur);
    // ---- optimiser stream state: the tile parked in LDS, its next row group, the two row groups in flight ----
    uint32_t pend_base = PARK;  // per-lane byte offset of row group 0 of the pending tile inside W / exp_avg / exp_avg_sq
    bool pend_lane = false;     // this lane owns a full 16-byte group in the pending tile (else its stream accesses stay parked)
    int pend_g = 16;            // next row group of the pending tile to issue (16: none left)
    struct Slot {
        f32x4 p, m, v;  // W, exp_avg, exp_avg_sq of the row group in flight
        uint32_t off;   // its per-lane byte offset (PARK: none -- loads return 0 without a fetch, stores are dropped)
        int gi;         // its index: gradient = element gi >> 2 of the parked accumulators 4 (gi & 3) + f
        bool live;      // holds a row group (wave-uniform)
    } sl[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        asm volatile("" : "=v"(sl[k].p));
        asm volatile("" : "=v"(sl[k].m));
        asm volatile("" : "=v"(sl[k].v));
        sl[k].off = PARK;
        sl[k].gi = 0;
        sl[k].live = false;
    }
    auto opt_pick = [&](Slot& s_) {  // next row group of the pending tile; parked when there is none
        const bool act = pend_g < 16;
        s_.live = act;
        s_.gi = act ? pend_g : 0;
        s_.off = (act && pend_lane && !(GD_ADAMW_DBG & 2)) ? pend_base + (uint32_t)pend_g * ldc4 : PARK;
        pend_g += act ? 1 : 0;
    };
    auto opt_update_store = [&](Slot& s_) {  // s_.p / m / v have landed
        const float* sg = reinterpret_cast<const float*>(stash + 256 * (s_.gi & 3)) + (s_.gi >> 2);
        float gr[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) gr[f] = sg[256 * f];  // accumulator 4 e + f, element t: 64 lanes x 16 B apart
        f32x4 pn = s_.p, mn = s_.m, vn = s_.v;  // (the slot's registers themselves are only ever written by its loads)
#if !(GD_ADAMW_DBG & 1)  // (probe builds, tools/build_variant.sh: bit 0 = no arithmetic, bit 1 = every stream access parked)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = pn[k], mk = mn[k], vk = vn[k];
            gd_adam_elem(pk, gr[k], mk, vk, hy);
            pn[k] = pk;
            mn[k] = mk;
            vn[k] = vk;
        }
#else
        pn[0] += gr[0] + gr[1] + gr[2] + gr[3];
#endif
        // (probe bit 3: stores parked; bit 4: stores go to lines the stream has NOT just loaded -- 448 rows further down)
        const uint32_t so = (GD_ADAMW_DBG & 8) ? PARK : ((GD_ADAMW_DBG & 16) && s_.off != PARK) ? s_.off + 448u * ldc4 : s_.off;
        dr_store(pn, srdW, so);
        dr_store(mn, srdM, so);
        dr_store(vn, srdV, so);
    };
    auto opt_load = [&](Slot& s_) {
        const uint32_t lo = (GD_ADAMW_DBG & 4) ? PARK : s_.off;  // (probe bit 2: loads parked)
        dr_load0_rw<NTL>(s_.p, srdW, lo);
        dr_load0_rw<NTL>(s_.m, srdM, lo);
        dr_load0_rw<NTL>(s_.v, srdV, lo);
    };
    auto opt_pin = [&](Slot& s_) {
        asm volatile("" : "+v"(s_.p));
        asm volatile("" : "+v"(s_.m));
        asm volatile("" : "+v"(s_.v));
    };

    f32x4 ra[R], rb[R];
    // ring fill, with the parked optimiser instructions a previous round would have issued in step u + 1 (see the header)
#pragma unroll
    for (int u = 0; u < D; ++u) {
        ra[u] = dr_load(srdA, offA, ka);
        rb[u] = dr_load(srdB, offB, kb);
        ka += sa;
        kb += sb;
        --l_left;
        const int w = (u + 1) % R;
        if (w == UA || w == UB) {
            // (parked STORES stand in for the loads too: a parked load would still write its destination when it lands, and
            // nothing keeps hipcc from using those registers meanwhile)
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < XS / 3; ++k) {
                dr_store(z, srdW, PARK);
                dr_store(z, srdM, PARK);
                dr_store(z, srdV, PARK);
            }
        }
    }
    const int q1 = (KSP / R / 4) * R, q2 = (KSP / R / 2) * R, q3 = (KSP / R * 3 / 4) * R;
    const bool defer_all = KSP / R >= 9;  // ring rounds per tile: two row groups each, issued in rounds 0..7, consumed by round 8
    for (;;) {
        unsigned int* tctr = &g_dr_ticket[d.ctr][qx][0];
        unsigned int tick = visited < 8 ? dr_ticket_issue(tctr) : 0u;
        const bool drew = visited < 8;
        int nxt = 0;
        const int tm = d.m_fastest ? (cur % d.tiles_m) : (cur / d.tiles_n);
        const int tn = d.m_fastest ? (cur / d.tiles_m) : (cur % d.tiles_n);
        const int m0 = tm * 64, n0 = tn * 64;
        f32x4 acc[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[e][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(0);
        for (int s0 = 0; s0 < KSP; s0 += R) {
            if (s0 == q1) __builtin_amdgcn_s_setprio(1);
            else if (s0 == q2) __builtin_amdgcn_s_setprio(2);
            else if (s0 == q3) __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int v = (u + D) % R;
                // step s has landed; steps s+1 .. s+D-1 and the optimiser instructions of every other step of a round stay in flight
                if (u == UA || u == UB) dr_wait<LPS*(D - 1) + XR - XS>();
                else dr_wait<LPS*(D - 1) + XR>();
                asm volatile("" : "+v"(ra[u]));
                asm volatile("" : "+v"(rb[u]));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int e = i / 4, f = i % 4;
                    acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u][e], rb[u][f], acc[e][f], 0, 0, 0);
                    if (i == 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        ra[v] = dr_load(srdA, offA, ka);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (i == 5) {
                        __builtin_amdgcn_sched_barrier(0);
                        rb[v] = dr_load(srdB, offB, kb);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (i == 9) {
                        __builtin_amdgcn_sched_barrier(0);
                        ka += sa;
                        kb += sb;
                        if (--l_left == 0) {  // once per tile: the cursor moves on to the next tile
                            nxt = -1;
                            if (drew) {
                                if (KSP < 2 * D + 2) dr_wait<0>();
                                asm volatile("" : "+v"(tick));
                                const int tk = __builtin_amdgcn_readfirstlane(tick);
                                const int nq = q_tiles(qx);
                                if (tk == nq + n_waves - 1 && lane == 0) __hip_atomic_store(tctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (tk < nq) {
                                    nxt = tile_of(qx, tk);
                                } else {
                                    qx = (qx + 1) & 7;
                                    ++visited;
                                    nxt = draw_blocking();
                                }
                            }
                            if (nxt < 0) nxt = ntiles;
                            set_cursor(nxt);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    } else if ((u == UA || u == UB) && i == 11) {
                        // ---- the optimiser stream's turn: the slot's row group (loaded one round ago) is updated and stored, and
                        // the slot reloaded with the next row group of the parked tile ----
                        __builtin_amdgcn_sched_barrier(0);
                        Slot& s_ = sl[u == UA ? 0 : 1];
                        // its loads are older than the ring loads of the R steps since (2 each) and the other slot's turn
                        dr_wait<LPS * R + XS>();
                        opt_pin(s_);
                        opt_update_store(s_);
                        opt_pick(s_);
                        opt_load(s_);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- end of tile ----
        const uint32_t vo = (uint32_t)(16 * q * g.ldc + n0 + 4 * r) * 4u;
        const int n = n0 + 4 * r;
        const bool lane_full = n + 3 < n_lim;  // the lane owns a full 16-byte group of every row of the tile
        if (defer_all) {
            // A k loop of >= 9 ring rounds has issued and consumed all sixteen row groups of the PREVIOUS tile (the slots hold
            // parked loads): park this one for the next k loop's stream.  Lanes without a full group (last column panel) stay
            // parked in the stream; what they own is updated element-wise below.
            // (The slots' registers are touched nowhere but in the k loop's turns: any other definition would make hipcc place
            // copies of them -- of registers in flight -- at the loop's back edge.)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int f = 0; f < 4; ++f) stash[64 * (4 * e + f)] = acc[e][f];
            pend_base = vo + (uint32_t)m0 * ldc4;
            pend_lane = lane_full;
            pend_g = 0;
        } else if (lane_full) {
            // a reduction too short for the stream (< 9 ring rounds): updated on the spot from the accumulators, as in round 3
            // (one memory round trip per row group)
            const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)c_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.aux), 0, (int)c_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.aux2), 0, (int)c_bytes, 0x00020000);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t o = vo + (uint32_t)(m0 + 4 * t + e) * ldc4;
                    f32x4 pv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, o, 0, 0));
                    f32x4 mv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rM, o, 0, 0));
                    f32x4 vv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rV, o, 0, 0));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float pk = pv[k], mk = mv[k], vk = vv[k];
                        gd_adam_elem(pk, acc[e][k][t], mk, vk, hy);
                        pv[k] = pk;
                        mv[k] = mk;
                        vv[k] = vk;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pv), rW, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mv), rM, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vv), rV, o, 0, 0);
                }
        }
        if (!lane_full && n < g.N) {
            // last column panel: the lane's group straddles the end of the row (N % 4 != 0) or holds the bias column -- element-wise
            // from the accumulators, now (at most one lane per row)
            float* __restrict__ Mo = const_cast<float*>(g.aux);
            float* __restrict__ Vo = const_cast<float*>(g.aux2);
            for (int t = 0; t < 4; ++t)
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + 16 * q + 4 * t + e;
                    if (m >= g.M) continue;
                    for (int k = 0; k < 4; ++k) {
                        if (n + k >= g.N) continue;
                        const int64_t o = (int64_t)m * g.ldc + n + k;
                        const float gk = acc[e][k][t];
                        if (g.out2 && n + k == g.N - 1) {  // the bias column (operand B's extra column): its own vector
                            g.out2[m] = gk;
                            continue;
                        }
                        float pk = g.C[o], mk = Mo[o], vk = Vo[o];
                        gd_adam_elem(pk, gk, mk, vk, hy);
                        g.C[o] = pk;
                        Mo[o] = mk;
                        Vo[o] = vk;
                    }
                }
        }
        if (nxt >= ntiles) break;
        cur = nxt;
    }
    // the parked cursor's loads and the stream's last instructions are still in flight: their registers stay live until they landed
    dr_wait<0>();
#pragma unroll
    for (int u = 0; u < R; ++u) {
        asm volatile("" ::"v"(ra[u]));
        asm volatile("" ::"v"(rb[u]));
    }
    // ---- the last tile's stream: nothing left to multiply.  What the last k loop left in the slots (landed), then the parked
    // tile's row groups four at a time -- twelve loads in flight per wave (everything asm has drained: hipcc schedules this) ----
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        opt_pin(sl[k]);
        opt_update_store(sl[k]);
    }
    dr_wait<0>();
    {
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)c_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.aux), 0, (int)c_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.aux2), 0, (int)c_bytes, 0x00020000);
        for (; pend_g < 16; pend_g += 4) {
            f32x4 pv[4], mv[4], vv[4];
            uint32_t oo[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                oo[u] = (pend_lane && !(GD_ADAMW_DBG & 2)) ? pend_base + (uint32_t)(pend_g + u) * ldc4 : PARK;
                pv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, oo[u], 0, 0));
                mv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rM, oo[u], 0, 0));
                vv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rV, oo[u], 0, 0));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int gi = pend_g + u;
                const float* sg = reinterpret_cast<const float*>(stash + 256 * (gi & 3)) + (gi >> 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float pk = pv[u][k], mk = mv[u][k], vk = vv[u][k];
                    gd_adam_elem(pk, sg[256 * k], mk, vk, hy);
                    pv[u][k] = pk;
                    mv[u][k] = mk;
                    vv[u][k] = vk;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pv[u]), rW, oo[u], 0, 2);  // (aux 2 = nt)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mv[u]), rM, oo[u], 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vv[u]), rV, oo[u], 0, 2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] * B[N,K]^T, both operands K-contiguous (the forward layers, reference models/DNN.py:79-86, with the fused
// row-loss / posterior / bias-activation epilogues of gaussian_diffusion.py:335, :451-498).
// A lane loads 16 bytes = four consecutive k of ONE row: lane (j = lane & 15, q = lane >> 4) reads P[row(j)][k0 + 4q .. +3],
// so a wave instruction covers 16 rows x 16 k and register component s is the operand of the MFMA that takes
// k in {k0 + s, k0 + 4 + s, k0 + 8 + s, k0 + 12 + s} (A and B permuted alike) -- four MFMA k-steps per load, the unit of the
// ring is therefore a CHUNK of 16 k.  Which matrix row a lane reads is free: A blocks take rows m0 + 16 i + j, B loads take
// rows n0 + 4 j + f (f = 0..3), which makes the accumulators
//     acc[i][f][t] = C[m0 + 16 i + 4 q + t][n0 + 4 r + f]
// -- again four consecutive columns per lane.  k past K lies inside the next row (no range check helps): the chunks that
// reach past K are masked with selects (one or two per tile).
// ---------------------------------------------------------------------------------------------------------------------
template <int TMB, int NB, int D, int EPI>
__global__ __launch_bounds__(512, 2) void dr_nt_kernel(const DrArgs d) {
    static_assert(EPI == GD_EPI_BIAS_ACT || EPI == GD_EPI_LOSS || EPI == GD_EPI_POST, "forward products");
    static_assert(NB == 4 || NB == 2, "B loads per chunk: 16 NB output columns, NB consecutive ones per lane");
    constexpr int LPC = TMB + NB;   // loads per chunk
    constexpr int R = D + 1;
    constexpr int NM = 4 * TMB * NB;  // MFMAs per chunk
    constexpr int SP = NM / (LPC + 1);  // the LPC loads of chunk c + D ride behind MFMA 2, 2 + SP, ...; then the cursor step
    static_assert(SP >= 2 && SP * LPC + 1 < NM, "loads and cursor step must fall inside the chunk");
    static_assert(LPC * D <= 63, "vmcnt is a 6-bit counter");
    const GdGemm& g = d.g;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int ntiles = d.tiles_m * d.tiles_n;
    unsigned int* ctr = &g_dr_ticket[d.ctr][0][0];
    const int n_waves = gridDim.x * 8;
    int cur = __builtin_amdgcn_readfirstlane(blockIdx.x * 8 + (threadIdx.x >> 6));
    if (cur >= ntiles) return;
    if (d.stagger > 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256)
        for (int i = 0; i < d.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    const int NCP = d.ksp;  // chunks run per tile (a multiple of R)
    const i32x4 srdA = dr_srd(g.A, (uint32_t)(((int64_t)(g.M - 1) * g.lda + g.K) * 4));
    const i32x4 srdB = dr_srd(g.B, (uint32_t)(((int64_t)(g.N - 1) * g.ldb + g.K) * 4));

    uint32_t offA[TMB], offB[NB];
    uint32_t kc = 0;  // byte offset of the chunk to load next (soffset, the same for both operands)
    int l_left = NCP;
    auto set_cursor = [&](int tile) {
        const bool ok = tile < ntiles;
        const int tm = tile % d.tiles_m, tn = tile / d.tiles_m;  // the row tiles of one column panel draw consecutive tickets
#pragma unroll
        for (int i = 0; i < TMB; ++i) offA[i] = ok ? (uint32_t)((tm * (16 * TMB) + 16 * i + r) * g.lda + 4 * q) * 4u : 0xFFFFFFF0u;
#pragma unroll
        for (int f = 0; f < NB; ++f) offB[f] = ok ? (uint32_t)((tn * (16 * NB) + NB * r + f) * g.ldb + 4 * q) * 4u : 0xFFFFFFF0u;
        kc = 0;
        l_left = NCP;
    };
    set_cursor(cur);
    f32x4 xa[R][TMB], xb[R][NB];
#pragma unroll
    for (int u = 0; u < D; ++u) {
#pragma unroll
        for (int i = 0; i < TMB; ++i) xa[u][i] = dr_load(srdA, offA[i], kc);
#pragma unroll
        for (int f = 0; f < NB; ++f) xb[u][f] = dr_load(srdB, offB[f], kc);
        kc += 64;
        --l_left;
    }
    const int q1 = (NCP / R / 4) * R, q2 = (NCP / R / 2) * R, q3 = (NCP / R * 3 / 4) * R;
    const int c_mask = g.K >> 4;  // first chunk that reaches past K (== chunks when K % 16 == 0: then only padded chunks)
    for (;;) {
        unsigned int tick = dr_ticket_issue(ctr);
        int nxt = 0;
        const int tm = cur % d.tiles_m, tn = cur / d.tiles_m;
        const int m0 = tm * 16 * TMB, n0 = tn * (16 * NB);
        f32x4 acc[TMB][NB];
#pragma unroll
        for (int i = 0; i < TMB; ++i)
#pragma unroll
            for (int f = 0; f < NB; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(0);
        for (int c0 = 0; c0 < NCP; c0 += R) {
            if (c0 == q1) __builtin_amdgcn_s_setprio(1);
            else if (c0 == q2) __builtin_amdgcn_s_setprio(2);
            else if (c0 == q3) __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int v = (u + D) % R;
                dr_wait<LPC*(D - 1)>();
#pragma unroll
                for (int i = 0; i < TMB; ++i) asm volatile("" : "+v"(xa[u][i]));
#pragma unroll
                for (int f = 0; f < NB; ++f) asm volatile("" : "+v"(xb[u][f]));
                if (c0 + u >= c_mask) {  // uniform; the last chunk(s) of a tile only: zero every k >= K
                    const int kq = (c0 + u) * 16 + 4 * q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool keep = kq + e < g.K;
#pragma unroll
                        for (int i = 0; i < TMB; ++i) xa[u][i][e] = keep ? xa[u][i][e] : 0.f;
#pragma unroll
                        for (int f = 0; f < NB; ++f) xb[u][f][e] = keep ? xb[u][f][e] : 0.f;
                    }
                }
#pragma unroll
                for (int n = 0; n < NM; ++n) {
                    const int e = n / (TMB * NB), i = (n / NB) % TMB, f = n % NB;
                    acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][i][e], xb[u][f][e], acc[i][f], 0, 0, 0);
                    if (n % SP == 1 && n / SP < LPC) {
                        __builtin_amdgcn_sched_barrier(0);
                        const int l = n / SP;
                        if (l < TMB) xa[v][l] = dr_load(srdA, offA[l], kc);
                        else xb[v][l - TMB] = dr_load(srdB, offB[l - TMB], kc);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (n == SP * LPC + 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        kc += 64;
                        if (--l_left == 0) {
                            if (NCP < 2 * D + 2) dr_wait<0>();
                            asm volatile("" : "+v"(tick));
                            const int tk = __builtin_amdgcn_readfirstlane(tick);
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
        // ---- epilogue.  acc[i][f][t] = C[m0 + 16 i + 4 q + t][n0 + NB r + f]: the lane owns NB consecutive columns of 4 TMB rows ----
        typedef float vecnb __attribute__((ext_vector_type(NB)));
        typedef vecnb vecnb_u __attribute__((aligned(4)));
        const int n = n0 + NB * r;
        const bool full = n + NB - 1 < g.N;
        float biasv[NB];
#pragma unroll
        for (int f = 0; f < NB; ++f) biasv[f] = g.bias ? g.bias[min(n + f, g.N - 1)] : 0.f;
        const bool has_z = (EPI == GD_EPI_POST) && (g.aux2 != nullptr);
        const bool has_r = (EPI == GD_EPI_POST) && (g.r2 != nullptr);
        auto put = [&](float* base, int64_t ld, int m, const float (&val)[NB]) {  // NB consecutive columns of row m (m < M)
            if (full) {
                vecnb w;
#pragma unroll
                for (int f = 0; f < NB; ++f) w[f] = val[f];
                *reinterpret_cast<vecnb_u*>(base + (int64_t)m * ld + n) = w;
            } else {
                for (int f = 0; f < NB; ++f)
                    if (n + f < g.N) base[(int64_t)m * ld + n + f] = val[f];
            }
        };
#pragma unroll
        for (int i = 0; i < TMB; ++i) {
            float racc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = m0 + 16 * i + 4 * q + t;
                const int mc = min(m, g.M - 1);
                const bool mok = m < g.M;
                float v4[NB];
#pragma unroll
                for (int f = 0; f < NB; ++f) v4[f] = acc[i][f][t] + biasv[f];
                racc[t] = 0.f;
                if (EPI == GD_EPI_BIAS_ACT) {
                    if (g.act == 1) {
#pragma unroll
                        for (int f = 0; f < NB; ++f) v4[f] = tanhf(v4[f]);
                    }
                    if (mok) put(g.C, g.ldc, m, v4);
                } else if (EPI == GD_EPI_LOSS) {
                    const float c1 = g.r0 ? g.r0[mc] : 1.f;
                    float tg[NB];
                    if (g.aux_bits) {
                        // {0,1} target rows as bitmaps: the lane's columns are NB bits of one word (n is a multiple of NB)
                        const uint32_t w = g.aux_bits[(int64_t)mc * g.ldbits + min((int64_t)(n >> 5), g.ldbits - 1)];
#pragma unroll
                        for (int f = 0; f < NB; ++f) tg[f] = (float)((w >> ((n & 31) + f)) & 1u);
                    } else {
#pragma unroll
                        for (int f = 0; f < NB; ++f) tg[f] = g.aux[(int64_t)mc * g.ldaux + min(n + f, g.N - 1)];
                    }
                    float dd[NB], ss = 0.f;
#pragma unroll
                    for (int f = 0; f < NB; ++f) {
                        dd[f] = c1 * v4[f] - tg[f];
                        if (mok && n + f < g.N) ss += dd[f] * dd[f];
                    }
                    if (mok) {
                        put(g.C, g.ldc, m, dd);
                        if (g.out2) put(g.out2, g.ldout2, m, v4);
                    }
                    racc[t] = ss;
                } else {  // GD_EPI_POST
                    const float c1 = g.r0[mc], c2 = g.r1[mc];
                    const float p1 = has_r ? g.r2[mc] : 0.f, p2 = has_r ? g.r3[mc] : 0.f;
                    const float sg = has_z ? g.r4[mc] : 0.f;
                    float pr[NB], mn[NB];
#pragma unroll
                    for (int f = 0; f < NB; ++f) {
                        const int nf = min(n + f, g.N - 1);
                        const float xt = g.aux[(int64_t)mc * g.ldaux + nf];
                        const float zz = has_z ? g.aux2[(int64_t)mc * g.ldaux2 + nf] : 0.f;
                        pr[f] = has_r ? (p1 * xt - p2 * v4[f]) : v4[f];
                        mn[f] = c1 * pr[f] + c2 * xt;
                        if (has_z) mn[f] += sg * zz;
                    }
                    if (mok) {
                        put(g.C, g.ldc, m, mn);
                        if (g.out2) put(g.out2, g.ldout2, m, pr);
                    }
                }
            }
            if (EPI == GD_EPI_LOSS) {
                // per-row sum of squares over the tile's columns: the 16 lanes r of a q-group hold one row
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float sv = racc[t];
                    sv += __shfl_xor(sv, 1);
                    sv += __shfl_xor(sv, 2);
                    sv += __shfl_xor(sv, 4);
                    sv += __shfl_xor(sv, 8);
                    const int m = m0 + 16 * i + 4 * q + t;
                    if (r == 0 && m < g.M) g.rowpart[(int64_t)m * g.ld_rowpart + tn] = sv;
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // one block of rows at a time: keeps the epilogue's live registers bounded
        }
        if (nxt >= ntiles) break;
        cur = nxt;
    }
    dr_wait<0>();
#pragma unroll
    for (int u = 0; u < R; ++u) {
#pragma unroll
        for (int i = 0; i < TMB; ++i) asm volatile("" ::"v"(xa[u][i]));
#pragma unroll
        for (int f = 0; f < NB; ++f) asm volatile("" ::"v"(xb[u][f]));
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] * B[N,K]^T for a SMALL A and a LARGE B (the output layer with the fused row loss, reference
// models/DNN.py:79-86 + gaussian_diffusion.py:335: A = hidden activations [batch, hidden], B = the [items, hidden] weight).
// What dr_nt_kernel showed: a K-contiguous operand must not be fetched straight into the MFMA layout (16 quarter lines per
// load).  Here
//   * A comes PRE-TRANSPOSED, At[K, M] (a 1.6 MB copy made by dr_transpose_kernel right before the launch), and is streamed
//     into registers exactly like dr_tn_kernel's operands: one dwordx4 load = rows m0 .. m0+63 of four k, plus one dword
//     load for rows m0+64 .. m0+79 -- an 80-row tile (batch 400 = 5 x 80);
//   * B is fetched in full 128-byte lines (8 lanes per row, 8 rows per load: "piece" u = rows 8u .. 8u+7 of a 32-k chunk),
//     written to a wave-PRIVATE LDS image (swizzled like gemm_f32.hip's K-contiguous image: conflict-free) and read back as
//     MFMA fragments with ds_read_b128.  Only the wave's own LDS queue orders the write and the read: no barrier, the eight
//     waves of a workgroup stay independent (2 x 8 KB per wave, 128 KB per workgroup).
// A fragment register component s of lane group q is k = 16 h + 4 q + s of its chunk half h, so A's lane group q loads k-row
// 4 q + s at step s: voffset carries 4 q rows, the scalar offset walks s = 0..3 and then jumps to the next half.
// One ring for everything, 8 steps (= one chunk) long: the loads of step sigma + 7 (A) and of piece u of chunk c + 2 (B) are
// issued between the MFMAs of step sigma = (c, u); the piece that has landed by then ((c+1, u+1), issued 7 steps earlier) is
// written to LDS at the start of the step; fragments are read at steps 3 (second half) and 7 (next chunk's first half).
//     acc[e][b][t] = C[m0 + 16 q + 4 t + e][n0 + 16 b + r]   (e < 4),      acc[4][b][t] = C[m0 + 64 + 4 q + t][n0 + 16 b + r].
// The ring is drained at the end of a tile (no operand load is in flight during the epilogue: nothing for hipcc to move);
// the SIMD partner owns the matrix pipe meanwhile.
// MEASURED (tools/gemm_probe, Yelp shape): correct and bit-identical to the LDS-tiled kernel, 0.262 ms against
// 0.2705 ms -- not the 0.22 ms hoped for, so it stays OPT-IN (GDMCF_GEMM_DR bit 3).  Why: (i) 2 690 tiles on 1 024 SIMDs are
// 2.63 tiles per SIMD: a third of the SIMDs run three tiles, the rest wait (the LDS-tiled kernel has the same 2.63 rounds);
// (ii) with the loop's loads parked outside the matrices (no memory traffic at all) the kernel is just as slow: the loop is
// bound by instruction issue -- a buffer_load blocks the SIMD's issue for ~50 cycles, 3 of them per 20 MFMAs (640 cycles) --
// and a second wave per SIMD does not fill those gaps (one wave per SIMD: 0.199 ms for 2 tiles each, two: 0.200 ms).
// ---------------------------------------------------------------------------------------------------------------------
// Fill loads of dr_hl_kernel: with its ~100 live scalars hipcc spills SGPRs into VGPR lanes and reloads a scalar offset with
// v_readlane right in front of the load that uses it.  An SGPR written by a VALU instruction must not be read by a
// vector-memory instruction for 5 wait states; hipcc pads its OWN instructions, it cannot see into an asm statement -- the
// load then uses the register's previous value (found the hard way: pieces of the first two chunks fetched from the
// previous load's offset).  The guarded forms wait inside the statement; build.py:lint_vmcnt rejects any unguarded case.
__device__ __forceinline__ f32x4 dr_load_g(i32x4 srd, uint32_t voff, uint32_t soff) {
    f32x4 v;
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(srd), "s"(soff) : "memory");
    return v;
}
__device__ __forceinline__ float dr_load1_g(i32x4 srd, uint32_t voff, uint32_t soff) {
    float v;
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(srd), "s"(soff) : "memory");
    return v;
}


__global__ __launch_bounds__(256) void dr_transpose_kernel(const float* __restrict__ A, int64_t lda, int M, int K,
                                                           float* __restrict__ At, int ldt) {
    __shared__ float t[32][33];
    const int k0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + y + 8 * i, k = k0 + x;
        t[y + 8 * i][x] = (m < M && k < K) ? A[(int64_t)m * lda + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + y + 8 * i, m = m0 + x;
        if (k < K && m < M) At[(int64_t)k * ldt + m] = t[x][y + 8 * i];
    }
}

// Loop loads of dr_hl_kernel as READ-WRITE operands: a slot then is live all the time, so hipcc cannot hand its registers to an
// accumulator between the ds_write that empties it and the load that refills it.  With "=v" outputs it did exactly that --
// every MFMA of the loop wrote its result into a different register quadruple than it read (v_mfma v[112:115], .., v[60:63]):
// legal, and measured no slower, but it mixes the accumulators into the ring and costs 22 registers (227 -> 205).
__device__ __forceinline__ void dr_load_rw(f32x4& v, i32x4 srd, uint32_t voff, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(v) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void dr_load1_rw(float& v, i32x4 srd, uint32_t voff, uint32_t soff) {
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(v) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
// probe ablations (tools/gemm_probe.hip): park the loop's A / B loads outside their matrices (they return 0 without a fetch)
#ifdef HL_NOA
#define HL_KA(x) (0x80000000u)
#else
#define HL_KA(x) (x)
#endif
#ifdef HL_NOB
#define HL_KB(x) (0x80000000u)
#else
#define HL_KB(x) (x)
#endif
template <int EPI>
__global__ __launch_bounds__(512, 2) void dr_hl_kernel(const DrArgs d) {
    static_assert(EPI == GD_EPI_LOSS, "output layer with the fused row loss");
    constexpr int LPS = 3;  // loads per step: A dwordx4, A dword, B dwordx4
    constexpr int D = 7;    // steps in flight beside the one being multiplied; ring = 8 steps = one 32-k chunk
    static_assert(LPS * D <= 63, "vmcnt is a 6-bit counter");
    const GdGemm& g = d.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) float dr_lds[];
    float* const lds = dr_lds + wave * 4096;  // 2 buffers x 64 rows x 32 k
    // float offsets inside a buffer.  Write: piece u = rows 8u + (lane >> 3), 16-byte slot (lane & 7) ^ ((row >> 1) & 7); the
    // swizzle term is (lane >> 4) for even u and 4 + (lane >> 4) for odd u.  Read: row 16 b + r, slot (4 h + q) ^ ((r >> 1) & 7).
    const int w_ev = (lane >> 3) * 32 + (((lane & 7) ^ (lane >> 4)) << 2), w_od = w_ev ^ 16;
    const int r_h0 = r * 32 + ((q ^ ((r >> 1) & 7)) << 2), r_h1 = r_h0 ^ 16;

    // ---- tiles and tickets: as in dr_tn_kernel (panel = the 80-row tiles of one 64-column slice of B) ----
    const int n_waves = gridDim.x * (blockDim.x >> 6);
    const int minor = d.tiles_m, panels = d.tiles_n;
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    int qx = xcc & 7, visited = 0;
    auto q_tiles = [&](int x) { return ((panels - x + 7) >> 3) * minor; };
    auto tile_of = [&](int x, int t) { return ((t / minor) * 8 + x) * minor + t % minor; };
    auto draw_blocking = [&]() {
        for (;;) {
            if (visited == 8) return -1;
            unsigned int* c = &g_dr_ticket[d.ctr][qx][0];
            unsigned int tk = dr_ticket_issue(c);
            dr_wait<0>();
            asm volatile("" : "+v"(tk));
            const int t = __builtin_amdgcn_readfirstlane(tk);
            const int n = q_tiles(qx);
            if (t == n + n_waves - 1 && lane == 0) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < n) return tile_of(qx, t);
            qx = (qx + 1) & 7;
            ++visited;
        }
    };
    if (d.stagger > 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256)  // (second wave of each SIMD; none in a 256-thread launch)
        for (int i = 0; i < d.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    int cur = draw_blocking();
    if (cur < 0) return;
    const int NCH = d.ksp;  // chunks per tile (even)
    const i32x4 srdA = dr_srd(g.A, (uint32_t)(((int64_t)(g.K - 1) * g.lda + g.M) * 4));  // At[K, M]
    const i32x4 srdB = dr_srd(g.B, (uint32_t)(((int64_t)(g.N - 1) * g.ldb + g.K) * 4));
    const uint32_t sa1 = 4u * (uint32_t)g.lda, sa13 = 13u * sa1;
    const uint32_t ub8 = 32u * (uint32_t)g.ldb;  // 8 rows of B
    const int q1 = (NCH / 8) * 2, q2 = (NCH / 4) * 2, q3 = (NCH * 3 / 8) * 2;

    for (;;) {
        unsigned int* tctr = &g_dr_ticket[d.ctr][qx][0];
        const bool drew = visited < 8;
        unsigned int tick = drew ? dr_ticket_issue(tctr) : 0u;
        const int tm = cur % d.tiles_m, tn = cur / d.tiles_m;
        const int m0 = tm * 80, n0 = tn * 64;
        const uint32_t oA4 = (uint32_t)(4 * q * g.lda + m0 + 4 * r) * 4u;
        const uint32_t oA1 = (uint32_t)(4 * q * g.lda + m0 + 64 + r) * 4u;
        const uint32_t oB = (uint32_t)(((int64_t)(n0 + (lane >> 3)) * g.ldb + 4 * (lane & 7)) * 4);
        uint32_t ka = 0;    // A: scalar offset of the next step to issue
        f32x4 ra4[8], G[8], FB[2][4];
        float ra1[8];
        f32x4 acc[5][4];
#pragma unroll
        for (int e = 0; e < 5; ++e)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[e][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(0);
        // ---- fill: chunk 0 of B through LDS buffer 0; then, in the order the steady state would have issued them, piece 0 of
        // chunk 1, and for u = 1..7 the A loads of step u - 1 and piece u of chunk 1 ----
#pragma unroll
        for (int u = 0; u < 8; ++u) G[u] = dr_load_g(srdB, oB, (uint32_t)u * ub8);
        dr_wait<0>();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("" : "+v"(G[u]));
            *reinterpret_cast<f32x4*>(lds + u * 256 + ((u & 1) ? w_od : w_ev)) = G[u];
        }
        G[0] = dr_load_g(srdB, oB, 128u);
#pragma unroll
        for (int u = 1; u < 8; ++u) {
            ra4[u - 1] = dr_load_g(srdA, oA4, ka);
            ra1[u - 1] = dr_load1_g(srdA, oA1, ka);
            ka += ((u - 1) & 3) == 3 ? sa13 : sa1;
            G[u] = dr_load_g(srdB, oB, 128u + (uint32_t)u * ub8);
        }
        dr_wait<21>();
        asm volatile("" : "=v"(ra4[7]));  // (slot 7 is first loaded by step 0: a defined value for its read-write operand)
        asm volatile("" : "=v"(ra1[7]));
        asm volatile("" : "+v"(G[0]));
        *reinterpret_cast<f32x4*>(lds + 2048 + w_ev) = G[0];
#pragma unroll
        for (int b = 0; b < 4; ++b) FB[0][b] = *reinterpret_cast<const f32x4*>(lds + b * 512 + r_h0);
        uint32_t kb = 256u;  // B: byte offset of chunk c + 2 inside a row

        for (int c0 = 0; c0 < NCH; c0 += 2) {
            if (c0 == q1) __builtin_amdgcn_s_setprio(1);
            else if (c0 == q2) __builtin_amdgcn_s_setprio(2);
            else if (c0 == q3) __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int uu = 0; uu < 16; ++uu) {
                const int p = uu >> 3, u = uu & 7, h = u >> 2, sx = u & 3;
                const int v = (u + 7) & 7;  // ring slot (= position inside its chunk) of step sigma + 7
                const int w = (u + 1) & 7;  // piece written to LDS in this step: (c+1, u+1), or (c+2, 0) at u = 7
                // chunks past the end of the tile are never multiplied: park their B loads outside the matrix (returns 0, no fetch)
                const uint32_t kbs = (c0 + p + 2 < NCH) ? kb : 0x80000000u;
                dr_wait<LPS * (D - 1)>();
                asm volatile("" : "+v"(ra4[u]));
                asm volatile("" : "+v"(ra1[u]));
                asm volatile("" : "+v"(G[w]));
                *reinterpret_cast<f32x4*>(lds + ((u < 7) ? (1 - p) : p) * 2048 + w * 256 + ((w & 1) ? w_od : w_ev)) = G[w];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 20; ++i) {
                    const int e = i >> 2, b = i & 3;
                    const float av = e < 4 ? ra4[u][e] : ra1[u];
                    acc[e][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, FB[h][b][sx], acc[e][b], 0, 0, 0);
                    if (i == 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        dr_load_rw(ra4[v], srdA, oA4, HL_KA(ka));
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (i == 5) {
                        __builtin_amdgcn_sched_barrier(0);
                        dr_load1_rw(ra1[v], srdA, oA1, HL_KA(ka));
                        ka += (v & 3) == 3 ? sa13 : sa1;
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (i == 9) {
                        __builtin_amdgcn_sched_barrier(0);
                        dr_load_rw(G[u], srdB, oB, HL_KB(kbs + (uint32_t)u * ub8));
                        if (u == 7) kb += 128u;
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (i == 13 && u == 3) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) FB[1][bb] = *reinterpret_cast<const f32x4*>(lds + p * 2048 + bb * 512 + r_h1);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (i == 13 && u == 7) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) FB[0][bb] = *reinterpret_cast<const f32x4*>(lds + (1 - p) * 2048 + bb * 512 + r_h0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- drain: whatever is still in flight belongs to steps past the tile; its registers stay untouched until it landed ----
        dr_wait<0>();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("" ::"v"(ra4[u]));
            asm volatile("" ::"v"(ra1[u]));
            asm volatile("" ::"v"(G[u]));
        }
        // the next tile (the ticket was issued before the fill: long landed)
        int nxt = -1;
        if (drew) {
            asm volatile("" : "+v"(tick));
            const int tk = __builtin_amdgcn_readfirstlane(tick);
            const int nq = q_tiles(qx);
            if (tk == nq + n_waves - 1 && lane == 0) __hip_atomic_store(tctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tk < nq) {
                nxt = tile_of(qx, tk);
            } else {
                qx = (qx + 1) & 7;
                ++visited;
                nxt = -2;  // draw after the epilogue (a few times per wave, at the end of the launch)
            }
        }
        // ---- epilogue (gaussian_diffusion.py:335): d = alpha * (acc + bias) - target, stored; per-row sum of d^2 over the tile ----
        {
            int ncl[4];
            bool nok[4];
            float biasv[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int n = n0 + 16 * b + r;
                nok[b] = n < g.N;
                ncl[b] = min(n, g.N - 1);
                biasv[b] = g.bias ? g.bias[ncl[b]] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 5; ++e) {
                float tg[4][4], c1v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = e < 4 ? m0 + 16 * q + 4 * t + e : m0 + 64 + 4 * q + t;
                    const int mc = min(m, g.M - 1);
                    c1v[t] = g.r0 ? g.r0[mc] : 1.f;
                    if (g.aux_bits) {
                        // {0,1} target rows as bitmaps: the tile's 64 columns are two words of the row (n0 is a multiple of 64)
                        uint32_t wv[2];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                            wv[jj] = g.aux_bits[(int64_t)mc * g.ldbits + min((int64_t)((n0 >> 5) + jj), g.ldbits - 1)];
#pragma unroll
                        for (int b = 0; b < 4; ++b) tg[t][b] = (float)((wv[b >> 1] >> (16 * (b & 1) + r)) & 1u);
                    } else {
#pragma unroll
                        for (int b = 0; b < 4; ++b) tg[t][b] = g.aux[(int64_t)mc * g.ldaux + ncl[b]];
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = e < 4 ? m0 + 16 * q + 4 * t + e : m0 + 64 + 4 * q + t;
                    const bool mok = m < g.M;
                    float ss = 0.f;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const float vv = acc[e][b][t] + biasv[b];
                        const float dd = c1v[t] * vv - tg[t][b];
                        if (mok && nok[b]) {
                            if (g.out2) g.out2[(int64_t)m * g.ldout2 + ncl[b]] = vv;
                            g.C[(int64_t)m * g.ldc + ncl[b]] = dd;
                            ss += dd * dd;
                        }
                    }
                    ss += __shfl_xor(ss, 1);
                    ss += __shfl_xor(ss, 2);
                    ss += __shfl_xor(ss, 4);
                    ss += __shfl_xor(ss, 8);
                    if (r == 0 && mok) g.rowpart[(int64_t)m * g.ld_rowpart + tn] = ss;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (nxt == -2) nxt = draw_blocking();
        if (nxt < 0) break;
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] * B[N,K]^T for a batch-sized A and a LARGE B with a fused row-loss / posterior epilogue (the output layer:
// reference models/DNN.py:83-86 with gaussian_diffusion.py:335 in training, :451-498 in the reverse loop), round 4:
// ONE FAT TILE PER WAVE, ONE WAVE PER SIMD, ONE PASS.
//
// What rounds 2-3 established about this product: the LDS-tiled kernel and the hybrid kernel both end at 0.6 of
// the matrix rate because (i) 1 345 / 2 690 tiles on 512 / 1 024 slots are 2.63 rounds -- a third of the chip runs three -- and
// (ii) the loop is bound by instruction ISSUE: v_mfma_f32_16x16x4_f32 shares the vector issue port, so every load, LDS access
// and barrier beside the MFMAs is matrix time lost (round 4: even another wave's arithmetic does not overlap).  Both have one
// cure: MORE OUTPUT PER WAVE.  A wave alone on its SIMD may use all 512 registers: 5 x NB accumulator blocks of 16 x 16 (NB =
// 11: 220 registers) hold an 80 x 176 tile, so the WHOLE [400 x 34 395] output is 980 tiles -- one per wave, 96 % of the 1 024
// SIMDs busy for the whole launch, no second round, no tail -- and per 16-deep k chunk the wave issues 220 MFMAs beside 16 loads,
// 11 LDS writes and 11 LDS reads (0.17 other instructions per MFMA; the LDS-tiled kernel: 0.35, plus a barrier per 80).
//   * A (the hidden activations, L2-resident) is loaded straight into the MFMA layout: lane (j = lane & 15, q = lane >> 4) reads
//     A[m0 + 16 i + j][k0 + 4 q .. + 3] -- one 16-byte load per row block and chunk; component s is the operand of the MFMA that
//     takes k = k0 + 4 q + s (A and B permuted alike).  1.6 MB read by every wave: half lines are no concern here (they were for
//     the STREAMED operand of dr_nt_kernel).
//   * B (the weight, streamed from HBM once per row tile: the five row tiles of a column panel run on one XCD) is fetched in
//     pieces of 16 rows x 64 B, staged in registers for two k steps, written to a wave-PRIVATE LDS image (rows of 16 floats,
//     16-byte slot s of row r at s ^ 2 ((r >> 2) & 1): conflict-free for the ds_read_b128 lane groups) and read back as
//     fragments -- ordered by the wave's own LDS queue, no barrier.
//   * No instruction of the loop is inline asm: every load of a chunk is waited for inside the chunk that issued it (nothing is
//     in flight across the loop's back edge), so hipcc's own counted waits are exact; sched_barriers pin the placement.
// acc[i][b][t] = C[m0 + 16 i + 4 q + t][n0 + 16 b + r].  Deterministic: fixed k order, one wave per tile, static assignment.
// ---------------------------------------------------------------------------------------------------------------------
template <int NB, int EPI>
__global__ __launch_bounds__(256, 1) void dr_fat_kernel(const DrArgs d) {
    static_assert(EPI == GD_EPI_LOSS || EPI == GD_EPI_POST, "output layer with a fused epilogue");
    static_assert(NB >= 4 && NB <= 12, "5 x NB accumulator blocks must fit 256 registers");
    constexpr int TMB = 5;
    const GdGemm& g = d.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) float dr_lds[];
    float* const lds = dr_lds + wave * (2 * NB * 256);  // two chunk images of 16 NB rows x 16 floats
    // consecutive tiles (the row tiles of one column panel first) on consecutive waves of ONE XCD: blocks b and b + 8 share an XCD
    const int nblk = gridDim.x, per = nblk >> 3;
    const int wl = ((nblk & 7) == 0 ? ((int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3)) : (int)blockIdx.x) * 4 + wave;
    const int n_waves = nblk * 4;
    const int ntiles = d.tiles_m * d.tiles_n;
    const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (int)(((int64_t)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (int)(((int64_t)(g.N - 1) * g.ldb + g.K) * 4), 0x00020000);
    const int NCH = d.ksp;       // chunks of 16 k per tile: ceil(K / 16) -- the loop runs them in pairs, an odd last one alone
    const int c_mask = g.K >> 4;  // first chunk that reaches past K
    // LDS float offsets: write -- piece p = rows 16 p + (lane >> 2), slot lane & 3; read -- block b = rows 16 b + r, slot q
    const int w_off = (lane >> 2) * 16 + (((lane & 3) ^ ((((lane >> 2) >> 2) & 1) << 1)) << 2);
    const int r_off = r * 16 + ((q ^ (((r >> 2) & 1) << 1)) << 2);
    for (int tile = wl; tile < ntiles; tile += n_waves) {
        const int tm = tile % d.tiles_m, tn = tile / d.tiles_m;
        const int m0 = tm * (16 * TMB), n0 = tn * (16 * NB);
        uint32_t offA[TMB], offB[NB];
#pragma unroll
        for (int i = 0; i < TMB; ++i) offA[i] = (uint32_t)((m0 + 16 * i + r) * g.lda + 4 * q) * 4u;
#pragma unroll
        for (int p = 0; p < NB; ++p) offB[p] = (uint32_t)(((int64_t)(n0 + 16 * p + (lane >> 2)) * g.ldb + 4 * (lane & 3)) * 4);
        f32x4 acc[TMB][NB];
#pragma unroll
        for (int i = 0; i < TMB; ++i)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[i][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 xa[2][TMB], G[NB], FB[NB];
        // ---- fill: chunk 0 of A into xa[0], chunk 0 of B through LDS image 0 ----
#pragma unroll
        for (int i = 0; i < TMB; ++i) xa[0][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdA, offA[i], 0, 0));
#pragma unroll
        for (int p = 0; p < NB; ++p) G[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdB, offB[p], 0, 0));
#pragma unroll
        for (int p = 0; p < NB; ++p) *reinterpret_cast<f32x4*>(lds + p * 256 + w_off) = G[p];

        // one chunk: PAR = its parity (A registers xa[PAR], LDS image PAR); the loads of chunk c + 1 ride in k steps 0 and 1,
        // its LDS writes in k step 3
#define GD_FAT_CHUNK(PAR, c)                                                                                          \
        {                                                                                                             \
            /* scalar offset of the next chunk; past the last chunk it parks the loads outside both matrices (0, no fetch) */ \
            const uint32_t kn = ((c) + 1 < NCH) ? (uint32_t)((c) + 1) * 64u : 0x80000000u;                            \
            _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                            \
                FB[b] = *reinterpret_cast<const f32x4*>(lds + (PAR) * (NB * 256) + b * 256 + r_off);                  \
            if ((c) >= c_mask) { /* the chunk(s) that reach past K: zero every k >= K of both operands (what lies behind a row's K */ \
                /* elements -- the next row, or the padding of a leading dimension > K -- may hold anything, NaN included) */ \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                       \
                    const bool keep = (c) * 16 + 4 * q + e < g.K;                                                     \
                    _Pragma("unroll") for (int i = 0; i < TMB; ++i) xa[PAR][i][e] = keep ? xa[PAR][i][e] : 0.f;       \
                    _Pragma("unroll") for (int b = 0; b < NB; ++b) FB[b][e] = keep ? FB[b][e] : 0.f;                  \
                }                                                                                                     \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            _Pragma("unroll") for (int sx = 0; sx < 4; ++sx) {                                                        \
                _Pragma("unroll") for (int n = 0; n < TMB * NB; ++n) {                                                \
                    const int i = n / NB, b = n % NB;                                                                 \
                    if (sx == 0 && i == 0) asm volatile("" : "+v"(FB[b])); /* (operands stay in VGPRs: hipcc otherwise parks them in spare AGPRs) */ \
                    if (sx == 0 && b == 0) asm volatile("" : "+v"(xa[PAR][i]));                                       \
                    acc[i][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[PAR][i][sx], FB[b][sx], acc[i][b], 0, 0, 0);  \
                    /* loads of the next chunk: A's five and the first pieces of B in k step 0, the rest in k step 1 */ \
                    if (sx < 2 && n % 5 == 2) { /* k step 0 carries loads 0 .. NB - 1, k step 1 the remaining TMB */     \
                        const int l = (sx == 0) ? n / 5 : NB + n / 5;                                                 \
                        if (l < TMB + NB) {                                                                           \
                            __builtin_amdgcn_sched_barrier(0);                                                        \
                            if (l < TMB)                                                                              \
                                xa[(PAR) ^ 1][l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(   \
                                    srdA, offA[l], kn, 0));                                                           \
                            else                                                                                      \
                                G[l - TMB] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(         \
                                    srdB, offB[l - TMB], kn, 0));                                                     \
                            __builtin_amdgcn_sched_barrier(0);                                                        \
                        }                                                                                             \
                    }                                                                                                 \
                    if (sx == 3 && n % 4 == 1 && n / 4 < NB) { /* the pieces have landed: into the other LDS image */  \
                        __builtin_amdgcn_sched_barrier(0);                                                            \
                        *reinterpret_cast<f32x4*>(lds + ((PAR) ^ 1) * (NB * 256) + (n / 4) * 256 + w_off) = G[n / 4]; \
                        __builtin_amdgcn_sched_barrier(0);                                                            \
                    }                                                                                                 \
                }                                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
            }                                                                                                         \
        }
        // Reverse step: the epilogue reads the tile's x_t (80 x 16 NB floats per wave, 55 MB per launch at the Yelp shape) in the same
        // burst in which every wave writes x_{t-1} -- the only HBM-bound stretch of the kernel, while the k loop leaves HBM nearly
        // idle.  Some chunks (default ten, ~30 us) before the end the wave touches one dword of every 128-byte line of its x_t tile with LDS-DMA
        // loads (buffer_load_dword ... lds into a scratch line of its own: no registers, nothing waits for them), so the epilogue's
        // reads find the lines in L2 / the Infinity Cache.
        const int c_pf = d.stagger > 0 ? ((NCH > d.stagger + 2 ? NCH - d.stagger : 0) & ~1) : -2;  // (d.stagger: chunks before the end; 0 = off)
        for (int c = 0; c + 1 < NCH; c += 2) {
            if constexpr (EPI == GD_EPI_POST) {
                if (c == c_pf) {
                    typedef __attribute__((address_space(3))) void* lds_vp;
                    constexpr int NJL = (16 * NB * 4 + 127) / 128 + 1;   // touches per row: one per line + the row's last element
                    constexpr int NPI = (80 * NJL + 63) / 64;
                    float* const scratch = dr_lds + 4 * (2 * NB * 256) + wave * 64;
                    const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float*>(g.aux), 0, (int)(((int64_t)(g.M - 1) * g.ldaux + g.N) * 4), 0x00020000);
#pragma unroll
                    for (int pi = 0; pi < NPI; ++pi) {
                        const int t = min(pi * 64 + lane, 80 * NJL - 1);
                        const int row = min(m0 + t % 80, g.M - 1), j = t / 80;
                        const int col = min(j < NJL - 1 ? n0 + 32 * j : n0 + 16 * NB - 1, g.N - 1);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdX, (lds_vp)scratch, 4, (int)(((int64_t)row * g.ldaux + col) * 4), 0, 0, 0);
                    }
                }
            }
            GD_FAT_CHUNK(0, c);
            GD_FAT_CHUNK(1, c + 1);
        }
        // an odd number of chunks (K = 1 000: 63, the last one half masked): the last chunk alone, not a pair with an all-zero
        // partner (1.6 % of the product's matrix instructions at K = 1 000)
        if (NCH & 1) GD_FAT_CHUNK(0, NCH - 1);
#undef GD_FAT_CHUNK

        // ---- epilogue: the tile goes through the wave's LDS (the chunk images are dead) one block of 16 rows at a time and leaves
        // in ROWS -- lane (r, q) owns four consecutive columns 4 (r + 16 j) of row 4 p + q: 16-byte accesses, 256 contiguous bytes
        // per row and instruction, 12 stores per row block instead of 44 (and as many target / x_t loads) ----
        constexpr int LDS_ = 16 * NB + 4;  // floats per staged row (+4: the four q groups of a ds_write_b32 hit different banks)
        constexpr int NJ = (4 * NB + 15) / 16;
        typedef f32x4 f32x4_e __attribute__((aligned(4)));
        f32x4 bias4[NJ];
        int col4[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            col4[j] = n0 + 4 * (r + 16 * j);
            bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (g.bias && r + 16 * j < 4 * NB) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bias4[j][e] = g.bias[min(col4[j] + e, g.N - 1)];
            }
        }
        const bool has_z = (EPI == GD_EPI_POST) && (g.aux2 != nullptr);
        const bool has_r = (EPI == GD_EPI_POST) && (g.r2 != nullptr);
        // What a step of four rows reads from global memory -- the rows' coefficients, their targets (LOSS) or x_t / noise (POST) --
        // is fetched ONE STEP AHEAD: the wave is alone on its SIMD, so a load waited for where it is issued stands still for a full
        // memory round trip, twenty times per tile (measured: the posterior product 0.272 ms in the reverse loop).  Rows are clamped
        // into the matrix, the 16-byte groups that do not lie inside it whole (last column tile) are fetched in their own step.
        struct Pre {
            f32x4 a[NJ], z[NJ];
            uint32_t w[NJ];
            float c1, c2, p1, p2, sg;
        } pre[2];
        auto fetch = [&](int i, int p4, Pre& P) {
            const int mc = min(m0 + 16 * i + 4 * p4 + q, g.M - 1);
            P.c1 = 1.f; P.c2 = 0.f; P.p1 = 0.f; P.p2 = 0.f; P.sg = 0.f;
            if (EPI == GD_EPI_LOSS) {
                if (g.r0) P.c1 = g.r0[mc];
            } else {
                P.c1 = g.r0[mc];
                P.c2 = g.r1[mc];
                if (has_r) { P.p1 = g.r2[mc]; P.p2 = g.r3[mc]; }
                if (has_z) P.sg = g.r4[mc];
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = col4[j];
                const bool whole = (r + 16 * j < 4 * NB) && n + 3 < g.N;  // (row clamped: the address is valid whatever m is)
                P.w[j] = 0u;
                P.a[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                P.z[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (EPI == GD_EPI_LOSS && g.aux_bits) {
                    P.w[j] = g.aux_bits[(int64_t)mc * g.ldbits + min((int64_t)(n >> 5), g.ldbits - 1)];
                } else if (whole) {
                    P.a[j] = *reinterpret_cast<const f32x4_e*>(g.aux + (int64_t)mc * g.ldaux + n);
                    if (has_z) P.z[j] = *reinterpret_cast<const f32x4_e*>(g.aux2 + (int64_t)mc * g.ldaux2 + n);
                }
            }
        };
        fetch(0, 0, pre[0]);
#pragma unroll
        for (int i = 0; i < TMB; ++i) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int t = 0; t < 4; ++t) lds[(4 * q + t) * LDS_ + 16 * b + r] = acc[i][b][t];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                const int step = 4 * i + p4;
                const Pre& P = pre[step & 1];
                if (step + 1 < 4 * TMB) fetch((step + 1) >> 2, (step + 1) & 3, pre[(step + 1) & 1]);
                const int m = m0 + 16 * i + 4 * p4 + q;
                const int mc = min(m, g.M - 1);
                const bool mok = m < g.M;
                float ss = 0.f;
                const float c1 = P.c1, c2 = P.c2, p1 = P.p1, p2 = P.p2, sg = P.sg;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int c4 = r + 16 * j;
                    const bool inb = c4 < 4 * NB;                      // inside the tile
                    const int n = col4[j];
                    const bool whole = inb && n + 3 < g.N;             // a whole 16-byte group inside the matrix's columns
                    const bool full = whole && mok;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(lds + (4 * p4 + q) * LDS_ + 4 * min(c4, 4 * NB - 1));
                    f32x4 o, o2 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (EPI == GD_EPI_LOSS) {
                        // d = alpha * (acc + bias) - target, stored; per-row sum of d^2 (gaussian_diffusion.py:335)
                        f32x4 tg;
                        if (g.aux_bits) {  // {0,1} target rows as bitmaps: four bits of one word (n is a multiple of 4)
                            const uint32_t w = P.w[j] >> (n & 31);
#pragma unroll
                            for (int e = 0; e < 4; ++e) tg[e] = (float)((w >> e) & 1u);
                        } else if (whole) {
                            tg = P.a[j];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) tg[e] = g.aux[(int64_t)mc * g.ldaux + min(n + e, g.N - 1)];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o2[e] = v[e] + bias4[j][e];
                            o[e] = c1 * o2[e] - tg[e];
                            if (inb && mok && n + e < g.N) ss += o[e] * o[e];
                        }
                    } else {
                        // posterior mean of the reverse step (gaussian_diffusion.py:451-471, :495-498, :210-217)
                        f32x4 xt = P.a[j], zz = P.z[j];
                        if (!whole) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                xt[e] = g.aux[(int64_t)mc * g.ldaux + min(n + e, g.N - 1)];
                                if (has_z) zz[e] = g.aux2[(int64_t)mc * g.ldaux2 + min(n + e, g.N - 1)];
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float vv = v[e] + bias4[j][e];
                            o2[e] = has_r ? (p1 * xt[e] - p2 * vv) : vv;  // pred_xstart
                            o[e] = c1 * o2[e] + c2 * xt[e];
                            if (has_z) o[e] += sg * zz[e];
                        }
                    }
                    if (full) {
                        *reinterpret_cast<f32x4_e*>(g.C + (int64_t)m * g.ldc + n) = o;
                        if (g.out2) *reinterpret_cast<f32x4_e*>(g.out2 + (int64_t)m * g.ldout2 + n) = o2;
                    } else if (inb && mok) {
                        for (int e = 0; e < 4 && n + e < g.N; ++e) {
                            g.C[(int64_t)m * g.ldc + n + e] = o[e];
                            if (g.out2) g.out2[(int64_t)m * g.ldout2 + n + e] = o2[e];
                        }
                    }
                }
                if (EPI == GD_EPI_LOSS) {
                    ss += __shfl_xor(ss, 1);
                    ss += __shfl_xor(ss, 2);
                    ss += __shfl_xor(ss, 4);
                    ss += __shfl_xor(ss, 8);
                    if (r == 0 && mok) g.rowpart[(int64_t)m * g.ld_rowpart + tn] = ss;
                }
                __builtin_amdgcn_sched_barrier(0);  // four rows at a time: keeps the epilogue's live registers bounded
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] * B[K,N] as split-K partial slabs, A K-contiguous, B row-contiguous along N (the input gradient
// dh = dZ * W of reference main.py:350 / models/DNN.py:83-86: A = dZ [batch, items], B = the output layer's weight [items, hidden]).
// Round 4, end: both operands go STRAIGHT into the MFMA register layout -- no LDS at all, no barrier, no inline asm:
//   A as in dr_fat_kernel: lane (r, q) loads A[m0 + 16 a + r][k0 + 4 q .. + 3], component s feeds the MFMA whose k slot q stands
//     for k0 + 4 q + s;
//   B as in dr_tn_kernel: load (s, l) brings rows k0 + 4 q + s, columns n0 + 64 l + 4 r .. + 3; register e is the operand of the
//     block whose sixteen columns are n0 + 64 l + 4 r + e -- so a lane ends with four CONSECUTIVE columns (e) per row.
// One wave per SIMD owns an 80 x (64 NL) tile over one K range (5 x 4 NL accumulator blocks): per 16-deep chunk 80 NL MFMAs beside
// 5 + 4 NL loads (0.08 other instructions per MFMA; the LDS-tiled kernel that served this product: 0.35 and a barrier per 80,
// MFMA pipe busy 0.72).  Tasks (split, tile) are dealt statically, the tiles of one split to one XCD (they share its rows of B).
// The slabs go to the same reducer as before (gd_splitk_reduce: row scale, tanh').  Deterministic: static assignment, fixed k order.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef GD_KN_EVERY
#define GD_KN_EVERY 4
#endif
template <int NL>
__global__ __launch_bounds__(256, 1) void dr_kn_kernel(const DrArgs d) {
    constexpr int TMB = 5;
    const GdGemm& g = d.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int nblk = gridDim.x, per = nblk >> 3;
    const int wl = ((nblk & 7) == 0 ? ((int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3)) : (int)blockIdx.x) * 4 + wave;
    const int n_waves = nblk * 4;
    const int ntiles = d.tiles_m * d.tiles_n;
    const int CPS = d.ksp;                      // chunks of 16 k per split (even)
    const int total_chunks = (g.K + 15) >> 4;
    const int ntasks = ntiles * g.splits;
    const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (int)(((int64_t)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    // (rows k >= K of B lie outside the descriptor and read as 0)
    const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (int)(((int64_t)(g.K - 1) * g.ldb + g.N) * 4), 0x00020000);
    const uint32_t ldb16 = (uint32_t)g.ldb * 64u;  // bytes per chunk of 16 rows of B
    for (int task = wl; task < ntasks; task += n_waves) {
        const int split = task / ntiles, tile = task - split * ntiles;
        const int tm = tile % d.tiles_m, tn = tile / d.tiles_m;
        const int m0 = tm * (16 * TMB), n0 = tn * (64 * NL);
        const int c_lo = split * CPS;
        uint32_t offA[TMB], offB[4][NL];
#pragma unroll
        for (int a = 0; a < TMB; ++a) offA[a] = (uint32_t)(((int64_t)min(m0 + 16 * a + r, g.M - 1) * g.lda + 4 * q) * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const int n = n0 + 64 * l + 4 * r;
                // a group that would start past the row's N columns is parked outside the matrix (reads 0)
                offB[s][l] = n < g.N ? (uint32_t)(((int64_t)(4 * q + s) * g.ldb + n) * 4) : 0x80000000u;
            }
        f32x4 acc[TMB][NL][4];
#pragma unroll
        for (int a = 0; a < TMB; ++a)
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[a][l][e] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 xa[2][TMB], xb[2][4][NL];
        {   // fill: chunk c_lo
            const uint32_t ka = (uint32_t)c_lo * 64u, kb = (uint32_t)c_lo * ldb16;
#pragma unroll
            for (int a = 0; a < TMB; ++a) xa[0][a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdA, offA[a], ka, 0));
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int l = 0; l < NL; ++l)
                    xb[0][s][l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdB, offB[s][l], kb, 0));
        }
        // one chunk: PAR = its parity; the 5 + 4 NL loads of chunk c + 1 ride between its first MFMAs, one every GD_KN_EVERY -- early, so
        // that the rest of the chunk covers their latency (B comes from HBM)
#define GD_KN_CHUNK(PAR, c)                                                                                            \
        {                                                                                                              \
            const bool more = (c) + 1 < c_lo + CPS && (c) + 1 < total_chunks;                                          \
            const uint32_t ka = more ? (uint32_t)((c) + 1) * 64u : 0x80000000u;                                        \
            const uint32_t kb = more ? (uint32_t)((c) + 1) * ldb16 : 0x80000000u;                                      \
            if ((c) * 16 + 15 >= g.K) { /* the chunk that reaches past K: zero A's k >= K (B's rows there read as 0) */ \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                        \
                    const bool keep = (c) * 16 + 4 * q + e < g.K;                                                      \
                    _Pragma("unroll") for (int a = 0; a < TMB; ++a) xa[PAR][a][e] = keep ? xa[PAR][a][e] : 0.f;        \
                }                                                                                                      \
            }                                                                                                          \
            /* (operands stay in VGPRs: hipcc otherwise parks them in spare AGPRs and moves them back per use) */        \
            _Pragma("unroll") for (int a = 0; a < TMB; ++a) asm volatile("" : "+v"(xa[PAR][a]));                       \
            _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                              \
                _Pragma("unroll") for (int l = 0; l < NL; ++l) asm volatile("" : "+v"(xb[PAR][s][l]));                 \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                            \
                _Pragma("unroll") for (int n = 0; n < TMB * NL * 4; ++n) {                                             \
                    const int a = n / (NL * 4), l = (n / 4) % NL, e = n % 4;                                           \
                    acc[a][l][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[PAR][a][s], xb[PAR][s][l][e], acc[a][l][e], 0, 0, 0); \
                    const int idx = s * (TMB * NL * 4) + n;                                                            \
                    if (idx % GD_KN_EVERY == 2 && idx / GD_KN_EVERY < TMB + 4 * NL) {                                  \
                        const int ld = idx / GD_KN_EVERY;                                                              \
                        __builtin_amdgcn_sched_barrier(0);                                                             \
                        if (ld < TMB)                                                                                  \
                            xa[(PAR) ^ 1][ld] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdA, offA[ld], ka, 0)); \
                        else                                                                                           \
                            xb[(PAR) ^ 1][(ld - TMB) / NL][(ld - TMB) % NL] = __builtin_bit_cast(                      \
                                f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdB, offB[(ld - TMB) / NL][(ld - TMB) % NL], kb, 0)); \
                        __builtin_amdgcn_sched_barrier(0);                                                             \
                    }                                                                                                  \
                }                                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
        }
        for (int c = c_lo; c < c_lo + CPS && c < total_chunks; c += 2) {
            GD_KN_CHUNK(0, c);
            GD_KN_CHUNK(1, c + 1);
        }
#undef GD_KN_CHUNK
        // ---- epilogue: the partial tile into slab `split`; lane (r, q) owns columns n0 + 64 l + 4 r .. + 3 of rows 16 a + 4 q + t ----
        float* __restrict__ slab = g.C + (int64_t)split * g.slab_stride;
#pragma unroll
        for (int a = 0; a < TMB; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = m0 + 16 * a + 4 * q + t;
                if (m >= g.M) continue;
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const int n = n0 + 64 * l + 4 * r;
                    const f32x4 v = {acc[a][l][0][t], acc[a][l][1][t], acc[a][l][2][t], acc[a][l][3][t]};
                    if (n + 3 < g.ldc) {  // (slab rows are round4(N) wide: a whole group or nothing)
                        *reinterpret_cast<f32x4*>(slab + (int64_t)m * g.ldc + n) = v;
                    } else {
                        for (int e = 0; e < 4; ++e)
                            if (n + e < g.N) slab[(int64_t)m * g.ldc + n + e] = v[e];
                    }
                }
            }
    }
}

template <int NB>
int dr_fat_go(const DrArgs& d, int epi, int n_cu, hipStream_t s) {
    const size_t lds = (size_t)4 * 2 * NB * 256 * sizeof(float) + 4 * 64 * sizeof(float);  // chunk images + a scratch line per wave (x_t prefetch)
    void (*kern)(const DrArgs) = epi == GD_EPI_LOSS ? dr_fat_kernel<NB, GD_EPI_LOSS> : dr_fat_kernel<NB, GD_EPI_POST>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[epi == GD_EPI_LOSS] && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            gdmcf_set_error("hipFuncSetAttribute(dr_fat_kernel, LDS=%zu): %s", lds, hipGetErrorString(e));
            return GDMCF_E_HIP;
        }
        attr_set[epi == GD_EPI_LOSS] = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_cu), dim3(256), lds, s, d);
    return GDMCF_OK;
}

void dr_hl_go(const DrArgs& d, int n_cu, hipStream_t s) {
    // GDMCF_HL_WAVES=4: one wave per SIMD (256 threads; 96 KB of LDS requested so that only one workgroup fits a CU)
    static const int waves = getenv("GDMCF_HL_WAVES") ? atoi(getenv("GDMCF_HL_WAVES")) : 8;
    if (waves == 4) hipLaunchKernelGGL((dr_hl_kernel<GD_EPI_LOSS>), dim3(n_cu), dim3(256), 96 * 1024, s, d);
    else hipLaunchKernelGGL((dr_hl_kernel<GD_EPI_LOSS>), dim3(n_cu), dim3(512), 128 * 1024, s, d);
}

// tile = 16 TMB rows x 16 NB columns; D chunks in flight beside the one being multiplied (ring of D + 1 slots)
template <int TMB, int NB, int D, int EPI>
void dr_nt_go(const DrArgs& d, int n_cu, hipStream_t s) {
    hipLaunchKernelGGL((dr_nt_kernel<TMB, NB, D, EPI>), dim3(n_cu), dim3(512), 0, s, d);
}

int dr_cu_count_fwd();
template <int D, bool NTL>
int dr_tn_adamw_go(const DrArgs& d, hipStream_t s) {
    static bool attr_set = false;  // 8 waves x 16 KB: the tile whose optimiser stream is running
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dr_tn_adamw_kernel<D, NTL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) {
            gdmcf_set_error("hipFuncSetAttribute(dr_tn_adamw_kernel, LDS=128 KB): %s", hipGetErrorString(e));
            return GDMCF_E_HIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((dr_tn_adamw_kernel<D, NTL>), dim3(dr_cu_count_fwd()), dim3(512), 128 * 1024, s, d);
    return GDMCF_OK;
}

template <int D, int EPI>
int dr_tn_go(const DrArgs& d, hipStream_t s) {
    if constexpr (EPI == GD_EPI_ADAMW) {
        // rows of W / exp_avg / exp_avg_sq on 128-byte lines: the stream's loads need not stay in L2 (dr_load0_rw)
        const GdGemm& g = d.g;
        const bool lines = (g.ldc & 31) == 0 && (((uintptr_t)g.C | (uintptr_t)g.aux | (uintptr_t)g.aux2) & 127) == 0;
        static const int force = getenv("GDMCF_DR_NT_LOADS") ? atoi(getenv("GDMCF_DR_NT_LOADS")) : -1;  // tuning knob: 0 / 1
        return (force >= 0 ? force != 0 : lines) ? dr_tn_adamw_go<D, true>(d, s) : dr_tn_adamw_go<D, false>(d, s);
    } else {
        hipLaunchKernelGGL((dr_tn_kernel<1, 1, D, EPI>), dim3(dr_cu_count_fwd()), dim3(512), 0, s, d);
    }
    return GDMCF_OK;
}

}  // namespace

static int dr_cu_count();
namespace { int dr_cu_count_fwd() { return dr_cu_count(); } }
static int dr_cu_count() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return n_cu;
}

// dr_kn_kernel: 80 x 128 tiles, one (split, tile) task per wave slot (4 per CU): as many splits as fill the slots once -- or 0 when the
// product is not one the kernel takes (rows that do not tile by 80 within 12 %, a reduction too short to split, few slots).
int gd_dr_kn_splits(int M, int N, int K) {
    static const int on_env = getenv("GDMCF_GEMM_DR") ? atoi(getenv("GDMCF_GEMM_DR")) : 49;
    if (!(on_env & 32) || M < 16 || N < 64 || K < 4096) return 0;
    const long tiles = (long)gd_cdiv(M, 80) * gd_cdiv(N, 128);
    if ((long)gd_cdiv(M, 80) * 80 * 100 > (long)M * 112) return 0;  // 80-row tiles: at most 12 % padding
    if ((long)gd_cdiv(N, 128) * 128 * 100 > (long)N * 112) return 0;
    const long slots = 4L * dr_cu_count();
    if (tiles > slots) return 0;
    int splits = (int)(slots / tiles);
    const int total_chunks = gd_cdiv(K, 16);
    if (splits > total_chunks / 8) splits = total_chunks / 8;  // at least eight chunks per split
    if (splits < 2 || splits > 64) return splits > 64 ? 64 : 0;
    return splits;
}

// Ticket-counter set of one launch.  Two launches that overlap in time -- the two weight gradients of a step on two streams
// (GDMCF_GEMM_SIDE=1), two host threads, a replayed graph beside an eager step -- must not draw from the same counters, or each
// computes only a subset of its tiles.  The set therefore belongs to the LAUNCH, not to the call site: eager launches rotate
// through sets 0..15, launches recorded during a stream capture through 16..31 (a graph node keeps its set for every replay, so
// it must never meet an eager launch's).  Limits that follow: at most 16 eager launches of these kernels in flight at once, and
// at most 16 captured ones among all graphs that replay concurrently -- stream order and graph order serialise far below that.
#include <atomic>
static std::atomic<unsigned> g_dr_seq_eager{0}, g_dr_seq_graph{0};
static int dr_ticket_slot(hipStream_t s) {
    // GDMCF_DR_TICKET_SLOT=n pins every launch to set n: the behaviour before round 4, kept as the negative control of
    // tests/test_gpu_reentrancy.py (overlapping launches then share their queues and lose tiles)
    static const int pinned = getenv("GDMCF_DR_TICKET_SLOT") ? atoi(getenv("GDMCF_DR_TICKET_SLOT")) : -1;
    if (pinned >= 0) return pinned & 31;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();  // (the legacy stream while another stream captures: not a capture of this launch)
        st = hipStreamCaptureStatusNone;
    }
    if (st == hipStreamCaptureStatusActive) return 16 + (int)(g_dr_seq_graph.fetch_add(1, std::memory_order_relaxed) & 15u);
    return (int)(g_dr_seq_eager.fetch_add(1, std::memory_order_relaxed) & 15u);
}

// Returns GD_DR_NOT_TAKEN when the product is not one this file handles (the caller falls back to the LDS-tiled kernels).
int g_gd_dr_force = -1;  // tools/gemm_probe.hip: overrides GDMCF_GEMM_DR per call when >= 0
int gd_gemm_dr_launch(int layA, int layB, int epi, GdGemm& g, hipStream_t s) {
    static const int on_env = getenv("GDMCF_GEMM_DR") ? atoi(getenv("GDMCF_GEMM_DR")) : 49;
    const int on = g_gd_dr_force >= 0 ? g_gd_dr_force : on_env;  // bit 0: weight gradients (default), bit 1: forward
    // products (opt-in: measured SLOWER than the LDS-tiled kernels -- 0.279 vs 0.270 ms for the Yelp loss product: a K-contiguous
    // operand costs 16 half-line L1 accesses per load instead of 8 full lines, TCP accesses x3.6, 20 % of the wave cycles waiting)
    if (!on || g.bf16) return GD_DR_NOT_TAKEN;
    if (g.accumulate || g.C16 || (g.splits > 1 && epi != GD_EPI_SLAB)) return GD_DR_NOT_TAKEN;  // (slabs: dr_kn_kernel sets its own split count)
    static const int stagger = getenv("GDMCF_DR_STAGGER") ? atoi(getenv("GDMCF_DR_STAGGER")) : 3;
    const int64_t lim = (int64_t)1 << 32;  // 32-bit byte offsets inside every matrix
    DrArgs d = {};
    d.ctr = -1;  // drawn per launch, once the product is known to be taken (dr_ticket_slot)
    d.stagger = stagger;
    // (the fused-AdamW epilogue: hipcc rotates accumulators through ring slots there, which the first, set-based lint
    // (build.py:lint_ring_registers) cannot tell from a copy of in-flight data; the per-register analysis that replaced it for
    // this variant (lint_vmcnt: no instruction touches a register whose load the counted waits do not cover) verifies it clean,
    // and tests/test_gpu_fullsize.py checks every element of W / exp_avg / exp_avg_sq at the full shapes.  GDMCF_GEMM_DR bit 2
    // clear (e.g. =1 with bit 2 masked by GDMCF_DR_NO_FUSED=1) sends the fused products back to the LDS-tiled kernel.)
    static const int no_fused = getenv("GDMCF_DR_NO_FUSED") ? atoi(getenv("GDMCF_DR_NO_FUSED")) : 0;
    if ((on & 1) && layA == GD_LAY_MC && layB == GD_LAY_MC && (epi == GD_EPI_STORE || (epi == GD_EPI_ADAMW && !no_fused))) {
        if ((int64_t)g.K * g.lda * 4 >= lim || (int64_t)g.K * g.ldb * 4 >= lim || (int64_t)g.M * g.ldc * 4 >= lim) return GD_DR_NOT_TAKEN;
        if (g.lda < g.M || g.ldb < g.N || g.ldc < g.N) return GD_DR_NOT_TAKEN;
        const long tiles = (long)gd_cdiv(g.M, 64) * gd_cdiv(g.N, 64);
        if (tiles < 512 || g.K < 128) return GD_DR_NOT_TAKEN;  // (short reductions: a tile is all prologue; the LDS-tiled kernels take them)
        // bias gradient requested as one more column of the product (linear.hip: operand B carries the row scale in column N): the
        // kernel multiplies N + 1 columns -- the extra one needs its lane's 4-column group to straddle the end, i.e. N % 4 == 0 or
        // any N (the straddle path stores element-wise) -- and writes it to out2 instead of C
        float* const bias_db = (g.ldb > g.N) ? g.out2 : nullptr;
        const int n_user = g.N;
        if (bias_db) g.N = n_user + 1;
        d.tiles_m = gd_cdiv(g.M, 64);
        d.tiles_n = gd_cdiv(g.N, 64);
        d.m_fastest = d.tiles_m <= d.tiles_n;  // tiles that share the LARGER operand's panel draw consecutive tickets
        {   // tuning knob: GDMCF_DR_MF=0|1 forces the ticket order of the fused-AdamW product
            static const int mf = getenv("GDMCF_DR_MF") ? atoi(getenv("GDMCF_DR_MF")) : -1;
            if (mf >= 0 && epi == GD_EPI_ADAMW) d.m_fastest = mf;
        }
        const int ks = gd_cdiv(g.K, 4);
        // ring depth: the one whose size wastes the fewest padded steps per tile
        int best = 9, waste = 1 << 30;
        for (int dd : {9, 8, 7}) {
            const int w = gd_cdiv(ks, dd + 1) * (dd + 1) - ks;
            if (w < waste) { waste = w; best = dd; }
        }
        {   // tuning knob: GDMCF_DR_D=7|8|9 forces the ring depth
            static const int forced = getenv("GDMCF_DR_D") ? atoi(getenv("GDMCF_DR_D")) : 0;
            if (forced >= 7 && forced <= 9) { best = forced; waste = gd_cdiv(ks, best + 1) * (best + 1) - ks; }
        }
        d.ksp = ks + waste;
        g.tiles_m = d.tiles_m;
        g.tiles_n = d.tiles_n;
        d.g = g;
        d.ctr = dr_ticket_slot(s);
        d.adam_dev = epi == GD_EPI_ADAMW ? g.adam_dev : nullptr;  // a bound graph step state (linear.hip)
        d.g.out2 = bias_db;
        {
            GdProfScope prof(g.prof_tag, 2.0 * g.M * n_user * g.K, s);
#define GD_DR_GO(DD)                                                      \
    do {                                                                  \
        int rc_ = (epi == GD_EPI_STORE) ? dr_tn_go<DD, GD_EPI_STORE>(d, s) : dr_tn_go<DD, GD_EPI_ADAMW>(d, s); \
        if (rc_ != GDMCF_OK) return rc_;                                  \
    } while (0)
            if (best == 9) GD_DR_GO(9);
            else if (best == 8) GD_DR_GO(8);
            else GD_DR_GO(7);
#undef GD_DR_GO
        }
        g.N = n_user;
        t_gd_last_gemm = epi == GD_EPI_ADAMW ? 3 : 2;
        if (bias_db) g.out2 = nullptr;  // taken: the caller skips its column-sum pass
        return gd_launch_status("gemm_dr");
    }
    // bit 5 (default on): the input gradient, split-K slabs, both operands straight into registers (dr_kn_kernel)
    if ((on & 32) && layA == GD_LAY_KC && layB == GD_LAY_MC && epi == GD_EPI_SLAB) {
        const int splits = gd_dr_kn_splits(g.M, g.N, g.K);
        const size_t need = (size_t)splits * g.M * g.ldc * sizeof(float);
        if (splits > 0 && g.ldc >= g.N && (g.ldc & 3) == 0 && g.ws_cap >= need && g.slab_stride >= (int64_t)g.M * g.ldc &&
            (int64_t)g.M * g.lda * 4 < ((int64_t)1 << 31) && (int64_t)g.K * g.ldb * 4 < ((int64_t)1 << 31) && g.lda >= g.K && g.ldb >= g.N &&
            (reinterpret_cast<uintptr_t>(g.B) & 15) == 0 && (g.ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0) {
            const int n_cu = dr_cu_count();
            const int total_chunks = gd_cdiv(g.K, 16);
            const int cps = (gd_cdiv(total_chunks, splits) + 1) & ~1;  // chunks per split, even (the loop runs them in pairs)
            d.tiles_m = gd_cdiv(g.M, 80);
            d.tiles_n = gd_cdiv(g.N, 128);
            d.ksp = cps;
            g.splits = gd_cdiv(total_chunks, cps);
            g.kchunk = cps * 16;
            g.tiles_m = d.tiles_m;
            g.tiles_n = d.tiles_n;
            d.g = g;
            {
                GdProfScope prof(g.prof_tag, 2.0 * g.M * g.N * g.K, s);
                hipLaunchKernelGGL((dr_kn_kernel<2>), dim3(n_cu), dim3(256), 0, s, d);
            }
            t_gd_last_gemm = 5;
            return gd_launch_status("gemm_dr_kn");
        }
    }
    // bit 4 (default on): the output layer with a fused epilogue as ONE FAT TILE PER WAVE (dr_fat_kernel)
    if ((on & 16) && layA == GD_LAY_KC && layB == GD_LAY_KC && (epi == GD_EPI_LOSS || epi == GD_EPI_POST)) {
        const int n_cu = dr_cu_count();
        const int tiles_m = gd_cdiv(g.M, 80);
        bool ok = (int64_t)g.M * g.lda * 4 < ((int64_t)1 << 31) && (int64_t)g.N * g.ldb * 4 < lim && g.lda >= g.K && g.ldb >= g.K &&
                  g.K >= 256 && (long)tiles_m * 80 * 100 <= (long)g.M * 112 &&  // 80-row tiles: at most 12 % padding
                  (n_cu & 7) == 0 && !(epi == GD_EPI_LOSS && g.rowpart == nullptr) &&
                  !(epi == GD_EPI_LOSS && g.aux_bits && g.ldbits < (g.N + 31) / 32);
        // width of the tile: the one whose rounds of one tile per SIMD cost the least matrix time (rounds x NB)
        int nb = 0;
        long best = 1L << 60;
        const long slots = 4L * n_cu;
        for (int c = 12; c >= 8 && ok; --c) {
            const long t = (long)tiles_m * gd_cdiv(g.N, 16 * c);
            const long cost = ((t + slots - 1) / slots) * c;
            if (t >= slots / 2 && cost < best) { best = cost; nb = c; }
        }
        if (ok && epi == GD_EPI_POST) {
            // the reverse step reads x_t and writes x_{t-1} in the epilogue (2 x 4 B per element): with every wave finishing at once
            // that burst runs under nothing, so this kernel only takes the product when the LDS-tiled kernel's last round of
            // workgroups would be badly filled (measured: Yelp width 0.248 against 0.267 ms, Amazon-Book width 0.706 against 0.663)
            const long t128 = (long)gd_cdiv(g.M, 80) * gd_cdiv(g.N, 128);
            const long rounds = (t128 + 2 * n_cu - 1) / (2 * n_cu);
            if (t128 * 100 >= rounds * 2 * n_cu * 90) ok = false;
        }
        if (ok && nb && !(epi == GD_EPI_LOSS && g.ld_rowpart < gd_cdiv(g.N, 16 * nb))) {
            d.tiles_m = tiles_m;
            d.tiles_n = gd_cdiv(g.N, 16 * nb);
            d.m_fastest = 1;
            d.ksp = (gd_cdiv(g.K, 16) + 1) & ~1;  // chunks of 16 k
            {
                static const int odd_on = getenv("GDMCF_FAT_ODD") ? atoi(getenv("GDMCF_FAT_ODD")) : 1;  // 0: the old even count (A/B)
                if (odd_on) d.ksp = gd_cdiv(g.K, 16);
                // x_t prefetch of the reverse step: this many chunks before the end of the k loop (0: off; A/B knob)
                static const int pf = getenv("GDMCF_FAT_PF") ? atoi(getenv("GDMCF_FAT_PF")) : 10;
                d.stagger = pf;  // (the field is unused by this kernel otherwise)
            }
            g.tiles_m = d.tiles_m;
            g.tiles_n = d.tiles_n;
            d.g = g;
            d.ctr = 0;
            int rc = GDMCF_OK;
            {
                GdProfScope prof(g.prof_tag, 2.0 * g.M * g.N * g.K, s);
                switch (nb) {
                    case 8: rc = dr_fat_go<8>(d, epi, n_cu, s); break;
                    case 9: rc = dr_fat_go<9>(d, epi, n_cu, s); break;
                    case 10: rc = dr_fat_go<10>(d, epi, n_cu, s); break;
                    case 11: rc = dr_fat_go<11>(d, epi, n_cu, s); break;
                    default: rc = dr_fat_go<12>(d, epi, n_cu, s); break;
                }
            }
            if (rc != GDMCF_OK) return rc;
            t_gd_last_gemm = 4;
            return gd_launch_status("gemm_dr");
        }
    }
    // bit 3: the output layer with the fused row loss on the hybrid kernel (A pre-transposed into the tail of the row-sum scratch)
    if ((on & 8) && layA == GD_LAY_KC && layB == GD_LAY_KC && epi == GD_EPI_LOSS) {
        if ((int64_t)g.N * g.ldb * 4 >= ((int64_t)1 << 31) || (int64_t)g.K * g.M * 4 >= lim) return GD_DR_NOT_TAKEN;
        if (g.lda < g.K || g.ldb < g.K || g.K < 256 || g.M < 64) return GD_DR_NOT_TAKEN;
        if (g.ldb != g.K && (g.K & 31)) return GD_DR_NOT_TAKEN;  // a chunk that reaches past K reads the next row: it must hold weights, not padding
        if ((long)gd_cdiv(g.M, 80) * 80 * 100 > (long)g.M * 112) return GD_DR_NOT_TAKEN;  // 80-row tiles: at most 12 % padding
        const int tiles_m = gd_cdiv(g.M, 80), tiles_n = gd_cdiv(g.N, 64);
        if ((long)tiles_m * tiles_n < 1024) return GD_DR_NOT_TAKEN;
        if (g.aux_bits && g.ldbits < (g.N + 31) / 32) return GD_DR_NOT_TAKEN;
        // scratch: the caller's row-sum buffer holds M x gdmcf_loss_tiles(N) floats; this kernel needs M x tiles_n of them
        const int64_t used = ((int64_t)g.M * tiles_n + 3) & ~(int64_t)3;
        if (g.rowpart == nullptr || used + (int64_t)g.K * g.M > (int64_t)g.M * g.ld_rowpart) return GD_DR_NOT_TAKEN;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dr_hl_kernel<GD_EPI_LOSS>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) {
                gdmcf_set_error("hipFuncSetAttribute(dr_hl_kernel, LDS=128 KB): %s", hipGetErrorString(e));
                return GDMCF_E_HIP;
            }
            attr_set = true;
        }
        float* at = g.rowpart + used;
        d.tiles_m = tiles_m;
        d.tiles_n = tiles_n;
        d.m_fastest = 1;
        d.ksp = (gd_cdiv(g.K, 32) + 1) & ~1;  // chunks of 32 k, an even number of them
        g.tiles_m = tiles_m;
        g.tiles_n = tiles_n;
        g.ld_rowpart = tiles_n;
        d.g = g;
        d.ctr = dr_ticket_slot(s);
        d.g.A = at;
        d.g.lda = g.M;
        {
            GdProfScope prof(g.prof_tag, 2.0 * g.M * g.N * g.K, s);
            hipLaunchKernelGGL(dr_transpose_kernel, dim3(gd_cdiv(g.K, 32), gd_cdiv(g.M, 32)), dim3(256), 0, s, g.A, g.lda, g.M, g.K, at, g.M);
            dr_hl_go(d, dr_cu_count(), s);
        }
        t_gd_last_gemm = 5;
        return gd_launch_status("gemm_dr");
    }
    if ((on & 2) && layA == GD_LAY_KC && layB == GD_LAY_KC && (epi == GD_EPI_LOSS || epi == GD_EPI_POST)) {
        if ((int64_t)g.M * g.lda * 4 >= lim || (int64_t)g.N * g.ldb * 4 >= lim) return GD_DR_NOT_TAKEN;
        if (g.lda < g.K || g.ldb < g.K || g.K < 256) return GD_DR_NOT_TAKEN;
        // batch-sized M in blocks of 16 rows: 5 blocks (80 rows) when that pads little (400 = 5 x 80), else 4
        const int tmb = ((long)gd_cdiv(g.M, 80) * 80 * 100 <= (long)gd_cdiv(g.M, 64) * 64 * 103) ? 5 : 4;
        // 64-column tiles halve the operand traffic per FLOP, 32-column tiles quantise better over 1 024 SIMDs (Yelp loss product:
        // 2 690 tiles of 80 x 64 = 2.6 per SIMD against 5 375 of 80 x 32)
        static const int nb_env = getenv("GDMCF_DR_NB") ? atoi(getenv("GDMCF_DR_NB")) : 0;
        const int n_cu = dr_cu_count();
        int nb = 4;
        {
            const long t4 = (long)gd_cdiv(g.M, 16 * tmb) * gd_cdiv(g.N, 64);
            const long per = (t4 + 4 * n_cu - 1) / (4 * n_cu);           // rounds of one tile per SIMD
            if (t4 * 100 < per * 4 * n_cu * 92 && per < 6) nb = 2;       // the last round would be < 92 % full
        }
        if (nb_env == 2 || nb_env == 4) nb = nb_env;
        d.tiles_m = gd_cdiv(g.M, 16 * tmb);
        d.tiles_n = gd_cdiv(g.N, 16 * nb);
        if ((long)d.tiles_m * d.tiles_n < 1024) return GD_DR_NOT_TAKEN;
        if (epi == GD_EPI_LOSS && (g.ld_rowpart < d.tiles_n || g.rowpart == nullptr)) return GD_DR_NOT_TAKEN;
        if (epi == GD_EPI_LOSS && g.aux_bits && g.ldbits < (g.N + 31) / 32) return GD_DR_NOT_TAKEN;
        d.m_fastest = 1;
        const int nc = gd_cdiv(g.K, 16);
        const int ring = nb == 4 ? 2 : 3;
        d.ksp = gd_cdiv(nc, ring) * ring;  // a multiple of the ring size
        g.tiles_m = d.tiles_m;
        g.tiles_n = d.tiles_n;
        d.g = g;
        d.ctr = dr_ticket_slot(s);
        {
            GdProfScope prof(g.prof_tag, 2.0 * g.M * g.N * g.K, s);
#define GD_DR_NT(T, NBV, DV)                                                            \
    do {                                                                                \
        if (epi == GD_EPI_LOSS) dr_nt_go<T, NBV, DV, GD_EPI_LOSS>(d, n_cu, s);          \
        else dr_nt_go<T, NBV, DV, GD_EPI_POST>(d, n_cu, s);                             \
    } while (0)
            if (tmb == 5 && nb == 4) GD_DR_NT(5, 4, 1);
            else if (tmb == 5) GD_DR_NT(5, 2, 2);
            else if (nb == 4) GD_DR_NT(4, 4, 1);
            else GD_DR_NT(4, 2, 2);
#undef GD_DR_NT
        }
        t_gd_last_gemm = 6;
        return gd_launch_status("gemm_dr");
    }
    return GD_DR_NOT_TAKEN;
}
