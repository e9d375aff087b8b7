// Probe (not part of the library): "direct-to-register" f32 MFMA GEMM -- no LDS, no barriers, independent persistent waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dr_probe.hip -o tools/bin/dr_probe
// The f32 matrix pipe (v_mfma_f32_16x16x4_f32) is 16x slower than the bf16 one, so an operand byte is worth 16x more MFMA
// time: a wave can afford to fetch its own operands from L1/L2 straight into the MFMA register layout.  For a
// row-contiguous operand P[k][rows] one global_load_dwordx4 per wave brings rows r0..r0+63 of four consecutive k:
// lane (i = lane & 15, q = lane >> 4) holds P[k0 + q][r0 + 4 i + e], e = 0..3 -- register e IS the operand of the MFMA block
// whose 16 rows are r0 + 4 i + e (any fixed assignment of matrix rows to MFMA rows is as good as another).  256 contiguous
// bytes per lane group, no transposition, and the result comes out as 16 contiguous bytes per lane along N.
// Waves are independent: each pulls 64x64 (or 128x64) output tiles off an atomic counter, keeps D k-steps of operands in
// flight (asm loads, counted vmcnt) and issues MFMAs back to back; co-resident waves fill each other's prologue/epilogue.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));

#define CK(x)                                                      \
    do {                                                           \
        hipError_t e_ = (x);                                       \
        if (e_ != hipSuccess) {                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

__device__ unsigned int g_ctr[16];

struct DrArgs {
    const float* A;  // [K][lda]   (row-contiguous: element (k, m) at A[k * lda + m])
    int64_t lda;
    const float* B;  // [K][ldb]
    int64_t ldb;
    float* C;        // [M][ldc]
    int64_t ldc;
    int M, N, K;
    int tiles_m, tiles_n, m_fastest;
    int ctr;
    int n_waves;
    int stagger;
    int prio;
    unsigned long long* stamps;  // [n_waves][6]: memtime start/end, memrealtime start/end, tiles, k-steps (diagnostic)
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor: loads whose offset falls outside [0, bytes) return 0 instead of faulting, so the pipeline may run past
// the end of K without any clamping arithmetic, and a k-step is advanced by ONE scalar add on the soffset operand
__device__ __forceinline__ i32x4 make_srd(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
template <int MODE>
__device__ __forceinline__ f32x4 gload(i32x4 srd, uint32_t voff, uint32_t soff) {
    f32x4 v;
    if (MODE & 4) asm volatile("" : "=v"(v) : "v"(voff), "s"(srd), "s"(soff));
    else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(srd), "s"(soff) : "memory");
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// TA / TB: 64-row load units per operand and k-step; D: k-steps in flight
// MODE bits (timing ablations): 1 = every tile loads tile 0's panels (cache resident), 2 = no result stores, 4 = no loads at all
// lane 0 draws a ticket; the returning atomic is issued as asm so that hipcc does not wait for it on the spot: it is the
// oldest operation of the tile and has long landed when the tile ends
__device__ __forceinline__ unsigned int ticket_issue(unsigned int* ctr) {
    unsigned int t;
    unsigned long long save;
    const unsigned int zero = 0, one = 1;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                 : "=&v"(t), "=&s"(save) : "v"(zero), "v"(one), "s"(ctr) : "memory");
    return t;
}

// WPS waves per SIMD; WPS == 2 runs as ONE 512-thread workgroup per CU: waves w and w + 4 share a SIMD; waves 4-7 start
// `stagger` x 3.4 us later so the pair is out of phase.
// The pipeline runs CONTINUOUSLY across tiles: while the last D steps of a tile are multiplied, the loads already belong to
// the next tile (its id comes from a ticket drawn one tile earlier), so a tile boundary costs neither a pipeline fill nor a
// drain.  The ring has R = D + 1 slots and every tile runs a multiple of R steps (steps past K load zeros through the
// buffer descriptor's range check), so slot indices stay compile-time constants.
template <int TA, int TB, int D, int WPS, int MODE>
__global__ __launch_bounds__(WPS == 2 ? 512 : 256, WPS) void dr_tn_kernel(const DrArgs g) {
    constexpr int LPS = TA + TB;  // loads per k-step
    constexpr int R = D + 1;
    static_assert(LPS * D <= 63, "vmcnt is a 6-bit counter");
    constexpr int WPB = WPS == 2 ? 8 : 4;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int ntiles = g.tiles_m * g.tiles_n;
    unsigned int* ctr = &g_ctr[g.ctr];
    const int n_waves = gridDim.x * WPB;
    int cur = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + (threadIdx.x >> 6));  // first tile: static; then n_waves + ticket
    if (WPS == 2 && g.stagger > 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256)
        for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    const int KS = g.K >> 2;               // k-steps (K % 4 == 0 in this probe)
    const int KSP = (KS + R - 1) / R * R;  // steps run per tile
    const i32x4 srdA = make_srd(g.A, (uint32_t)g.K * (uint32_t)g.lda * 4u), srdB = make_srd(g.B, (uint32_t)g.K * (uint32_t)g.ldb * 4u);
    const i32x4 srdC = make_srd(g.C, (uint32_t)g.M * (uint32_t)g.ldc * 4u);
    const uint32_t sa = 16u * (uint32_t)g.lda, sb = 16u * (uint32_t)g.ldb;
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    int st_tiles = 0;
    if (cur >= ntiles) return;

    // ---- load cursor: the tile whose operands are being fetched ----
    uint32_t offA[TA], offB[TB];  // per-lane byte offsets of the cursor's tile
    uint32_t ka = 0, kb = 0;      // byte offset of the k-step to load next (soffset operand)
    int l_left = KSP;             // steps of the cursor's tile not yet issued
    auto set_cursor = [&](int tile) {
        // a tile id past the end parks the cursor outside both buffers: every load returns zeros
        const bool ok = tile < ntiles;
        const int tm = g.m_fastest ? (tile % g.tiles_m) : (tile / g.tiles_n);
        const int tn = g.m_fastest ? (tile / g.tiles_m) : (tile % g.tiles_n);
        const int lm0 = (MODE & 1) ? 0 : tm * 64 * TA, ln0 = (MODE & 1) ? 0 : tn * 64 * TB;
#pragma unroll
        for (int a = 0; a < TA; ++a) offA[a] = ok ? (uint32_t)(q * g.lda + lm0 + 64 * a + 4 * r) * 4u : 0xFFFFFFF0u;
#pragma unroll
        for (int b = 0; b < TB; ++b) offB[b] = ok ? (uint32_t)(q * g.ldb + ln0 + 64 * b + 4 * r) * 4u : 0xFFFFFFF0u;
        ka = kb = 0;
        l_left = KSP;
    };
    set_cursor(cur);
    f32x4 ra[R][TA], rb[R][TB];
#define DR_ADVANCE()  \
    do {              \
        ka += sa;     \
        kb += sb;     \
        --l_left;     \
    } while (0)
#pragma unroll
    for (int u = 0; u < D; ++u) {
#pragma unroll
        for (int a = 0; a < TA; ++a) ra[u][a] = gload<MODE>(srdA, offA[a], ka);
#pragma unroll
        for (int b = 0; b < TB; ++b) rb[u][b] = gload<MODE>(srdB, offB[b], kb);
        DR_ADVANCE();
    }
    for (;;) {
        ++st_tiles;
        unsigned int tick = ticket_issue(ctr);  // id of the tile AFTER this one: needed when the cursor leaves this tile
        int nxt = 0;
        const int tm = g.m_fastest ? (cur % g.tiles_m) : (cur / g.tiles_n);
        const int tn = g.m_fastest ? (cur / g.tiles_m) : (cur % g.tiles_n);
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
        const int q1 = (KSP / R / 4) * R, q2 = (KSP / R / 2) * R, q3 = (KSP / R * 3 / 4) * R;
        if (g.prio) __builtin_amdgcn_s_setprio(0);
        for (int s0 = 0; s0 < KSP; s0 += R) {
            // two waves share a SIMD and the arbiter prefers the OLDER one: the younger is starved and its tile finishes late (a
            // long tail at the end of the launch).  Priority by progress: the wave closer to the end of its tile wins.
            if (g.prio) {
                if (s0 == q1) __builtin_amdgcn_s_setprio(1);
                else if (s0 == q2) __builtin_amdgcn_s_setprio(2);
                else if (s0 == q3) __builtin_amdgcn_s_setprio(3);
            }
#pragma unroll
            for (int u = 0; u < R; ++u) {
                constexpr int NM = 16 * TA * TB;  // MFMAs of this step; the LPS loads ride behind MFMA 2, 6, 10, ...
                const int v = (u + D) % R;        // slot of step s + D (= the slot step s - 1 has left)
                wait_vm<LPS*(D - 1)>();           // step s has landed; steps s+1 .. s+D-1 stay in flight
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
                        if (l < TA) ra[v][l] = gload<MODE>(srdA, offA[l], ka);
                        else rb[v][l - TA] = gload<MODE>(srdB, offB[l - TA], kb);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (i == 4 * LPS + 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        DR_ADVANCE();
                        if (l_left == 0) {  // once per tile: the cursor moves on to the next tile
                            asm volatile("" : "+v"(tick));  // the ticket is older than every load of the last D steps: landed
                            const int tk = __builtin_amdgcn_readfirstlane(tick);
                            if (tk == ntiles - 1 && lane == 0) atomicExch(ctr, 0u);  // ntiles tickets per launch: the last one resets
                            nxt = n_waves + tk;
                            set_cursor(nxt);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // epilogue: acc[a][e][b][f][t] = C[m0 + 64a + 16q + 4t + e][n0 + 64b + 4r + f]: 16 bytes per lane along N, rows past M
        // fall outside the descriptor; the row part of the address is scalar (soffset)
        if (!(MODE & 2)) {
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
                                const f32x4 v = {acc[a][e][b][0][t], acc[a][e][b][1][t], acc[a][e][b][2][t], acc[a][e][b][3][t]};
                                const uint32_t so = (uint32_t)(m0 + 64 * a + 4 * t + e) * (uint32_t)g.ldc * 4u;
                                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(vo), "s"(srdC), "s"(so) : "memory");
                            }
                } else if (n < g.N) {
                    for (int a = 0; a < TA; ++a)
                        for (int t = 0; t < 4; ++t)
                            for (int e = 0; e < 4; ++e) {
                                const int m = m0 + 64 * a + 16 * q + 4 * t + e;
                                if (m < g.M)
                                    for (int k = 0; k < 4; ++k)
                                        if (n + k < g.N) g.C[(int64_t)m * g.ldc + n + k] = acc[a][e][b][k][t];
                            }
                }
            }
        }
        if (nxt >= ntiles) break;
        cur = nxt;
    }
#undef DR_ADVANCE
    // the parked cursor's loads are still in flight: their registers stay live until they have landed
    wait_vm<0>();
#pragma unroll
    for (int u = 0; u < R; ++u) {
#pragma unroll
        for (int a = 0; a < TA; ++a) asm volatile("" ::"v"(ra[u][a]));
#pragma unroll
        for (int b = 0; b < TB; ++b) asm volatile("" ::"v"(rb[u][b]));
    }
    if (g.stamps && lane == 0) {
        unsigned long long* o = g.stamps + (size_t)(blockIdx.x * WPB + (threadIdx.x >> 6)) * 6;
        o[0] = st_c0; o[1] = __builtin_amdgcn_s_memtime(); o[2] = st_r0; o[3] = __builtin_amdgcn_s_memrealtime(); o[4] = st_tiles; o[5] = (unsigned long long)st_tiles * KS;
    }
}

template <int TA, int TB, int D, int WPS, int MODE = 0>
float run(DrArgs g, int reps, const char* name) {
    g.tiles_m = (g.M + 64 * TA - 1) / (64 * TA);
    g.tiles_n = (g.N + 64 * TB - 1) / (64 * TB);
    g.m_fastest = g.tiles_m <= g.tiles_n;
    const int blocks = WPS == 2 ? 256 : 256 * WPS;
    const int threads = WPS == 2 ? 512 : 256;
    g.n_waves = blocks * (threads / 64);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // warm-up long enough for the clock to settle (the first tens of ms after an idle period run ~10 % slower)
    for (int i = 0; i < 150; ++i) hipLaunchKernelGGL((dr_tn_kernel<TA, TB, D, WPS, MODE>), dim3(blocks), dim3(threads), 0, 0, g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((dr_tn_kernel<TA, TB, D, WPS, MODE>), dim3(blocks), dim3(threads), 0, 0, g);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    {
        unsigned long long* st;
        CK(hipMalloc(&st, (size_t)g.n_waves * 48));
        CK(hipMemset(st, 0, (size_t)g.n_waves * 48));
        DrArgs g2_ = g; g2_.stamps = st;
        hipLaunchKernelGGL((dr_tn_kernel<TA, TB, D, WPS, MODE>), dim3(blocks), dim3(threads), 0, 0, g2_);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)g.n_waves * 6);
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        CK(hipFree(st));
        std::vector<double> cps, clk; double tmax = 0, tmin = 1e30; int tl_min = 1 << 30, tl_max = 0;
        for (int w = 0; w < g.n_waves; ++w) {
            const unsigned long long* o = &h[(size_t)w * 6];
            if (!o[5]) continue;
            cps.push_back((double)(o[1] - o[0]) / (double)o[5]);
            clk.push_back((double)(o[1] - o[0]) / (double)(o[3] - o[2]) * 0.1);  // GHz: memrealtime ticks at 100 MHz
            tmax = std::max(tmax, (double)(o[3] - o[2]) * 0.01); tmin = std::min(tmin, (double)(o[3] - o[2]) * 0.01);
            tl_min = std::min(tl_min, (int)o[4]); tl_max = std::max(tl_max, (int)o[4]);
        }
        std::sort(cps.begin(), cps.end()); std::sort(clk.begin(), clk.end());
        if (!cps.empty()) printf("   per wave: median %.0f shader cycles per k-step (%d MFMAs = %d cycles of pipe), clock %.2f GHz, wave lifetime %.0f..%.0f us, tiles per wave %d..%d\n",
                                 cps[cps.size() / 2], 16 * TA * TB, 512 * TA * TB, clk[clk.size() / 2], tmin, tmax, tl_min, tl_max);
    }
    printf("%-10s prio %d stagger %d mode %d TA=%d TB=%d D=%d waves/SIMD=%d  %.4f ms  %.1f TF  (%d tiles)\n", name, g.prio, g.stagger, MODE, TA, TB, D, WPS, ms,
           2.0 * g.M * g.N * g.K / ms / 1e9, g.tiles_m * g.tiles_n);
    return ms;
}

static double check(const DrArgs& g, const std::vector<float>& hA, const std::vector<float>& hB, int samples) {
    std::vector<float> hC((size_t)g.M * g.ldc);
    CK(hipMemcpy(hC.data(), g.C, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    srand(7);
    for (int sidx = 0; sidx < samples; ++sidx) {
        int m = rand() % g.M, n = rand() % g.N;
        if (sidx < 64) { m = g.M - 1 - (sidx & 7); n = g.N - 1 - (sidx >> 3); }  // the edges
        double ref = 0, mag = 0;
        for (int k = 0; k < g.K; ++k) {
            const double p = (double)hA[(size_t)k * g.lda + m] * hB[(size_t)k * g.ldb + n];
            ref += p;
            mag += p < 0 ? -p : p;
        }
        const double err = fabs(hC[(size_t)m * g.ldc + n] - ref) / (mag + 1e-30);
        if (err > worst) worst = err;
    }
    return worst;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int B = 400, I = 34395, H = 1000, E = 10, ldi = 34432, ldh = 1024;
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    std::vector<float> hbig((size_t)B * ldi), hsmall((size_t)B * ldh);
    srand(1);
    for (auto& v : hbig) v = (rand() / (float)RAND_MAX - 0.5f);
    for (auto& v : hsmall) v = (rand() / (float)RAND_MAX - 0.5f);
    float *big, *small_, *dW2, *dW1;
    CK(hipMalloc(&big, hbig.size() * 4));
    CK(hipMalloc(&small_, hsmall.size() * 4));
    CK(hipMalloc(&dW2, (size_t)I * H * 4));
    CK(hipMalloc(&dW1, (size_t)H * (I + E) * 4));
    CK(hipMemcpy(big, hbig.data(), hbig.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(small_, hsmall.data(), hsmall.size() * 4, hipMemcpyHostToDevice));
    // dW2[I][H] = diff[B][I]^T * sh[B][H]
    DrArgs g2 = {};
    g2.A = big; g2.lda = ldi; g2.B = small_; g2.ldb = ldh; g2.C = dW2; g2.ldc = H; g2.M = I; g2.N = H; g2.K = B; g2.ctr = 5;
    // dW1[H][I+E] = dz[B][H]^T * xin[B][I+E]
    DrArgs g1 = {};
    g1.A = small_; g1.lda = ldh; g1.B = big; g1.ldb = ldi; g1.C = dW1; g1.ldc = I + E; g1.M = H; g1.N = I + E; g1.K = B; g1.ctr = 6;

#define BOTH(TA, TB, D, W)                                                                  \
    do {                                                                                    \
        CK(hipMemset(dW2, 0xff, (size_t)I * H * 4));                                        \
        CK(hipMemset(dW1, 0xff, (size_t)H * (I + E) * 4));                                  \
        run<TA, TB, D, W>(g2, reps, "dW2");                                                 \
        run<TA, TB, D, W>(g1, reps, "dW1");                                                 \
        printf("   max rel err (of sum |a b|): dW2 %.2e  dW1 %.2e\n", check(g2, hbig, hsmall, 4000), check(g1, hsmall, hbig, 4000)); \
    } while (0)
    g1.stagger = g2.stagger = 0;
    BOTH(1, 1, 9, 2);
    g1.prio = g2.prio = 1;
    BOTH(1, 1, 9, 2);
    BOTH(1, 1, 4, 2);
    g1.stagger = g2.stagger = 3;
    BOTH(1, 1, 9, 2);
    BOTH(1, 1, 4, 2);
    BOTH(2, 1, 4, 2);
    BOTH(1, 2, 4, 2);
    return 0;
}
