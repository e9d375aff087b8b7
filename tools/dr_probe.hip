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
};

template <int MODE>
__device__ __forceinline__ f32x4 gload(const float* base, uint32_t off) {
    f32x4 v;
    if (MODE & 4) asm volatile("" : "=v"(v) : "v"(off), "s"(base));
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// TA / TB: 64-row load units per operand and k-step; D: k-steps in flight
// MODE bits (timing ablations): 1 = every tile loads tile 0's panels (cache resident), 2 = no result stores, 4 = no loads at all
template <int TA, int TB, int D, int WPS, int MODE>
__global__ __launch_bounds__(256, WPS) void dr_tn_kernel(const DrArgs g) {
    constexpr int LPS = TA + TB;  // loads per k-step
    static_assert(LPS * (D - 1) <= 63, "vmcnt is a 6-bit counter");
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int ntiles = g.tiles_m * g.tiles_n;
    unsigned int* ctr = &g_ctr[g.ctr];

    int tile;
    {
        unsigned int t0 = 0;
        if (lane == 0) t0 = atomicAdd(ctr, 1u);
        tile = __builtin_amdgcn_readfirstlane(t0);
    }
    const int KS = g.K >> 2;  // k-steps (K % 4 == 0 in this probe)
    while (tile < ntiles) {
        unsigned int tn_ = 0;
        if (lane == 0) tn_ = atomicAdd(ctr, 1u);  // the next tile's id travels under this tile's work
        const int tm = g.m_fastest ? (tile % g.tiles_m) : (tile / g.tiles_n);
        const int tn = g.m_fastest ? (tile / g.tiles_m) : (tile % g.tiles_n);
        const int m0 = tm * 64 * TA, n0 = tn * 64 * TB;
        const int lm0 = (MODE & 1) ? 0 : m0, ln0 = (MODE & 1) ? 0 : n0;
        uint32_t offA[TA], offB[TB];
#pragma unroll
        for (int a = 0; a < TA; ++a) offA[a] = (uint32_t)(q * g.lda + min(lm0 + 64 * a + 4 * r, (int)g.lda - 4)) * 4u;
#pragma unroll
        for (int b = 0; b < TB; ++b) offB[b] = (uint32_t)(q * g.ldb + min(ln0 + 64 * b + 4 * r, (int)g.ldb - 4)) * 4u;
        const float* pa = g.A;
        const float* pb = g.B;
        const int64_t sa = 4 * g.lda, sb = 4 * g.ldb;

        f32x4 acc[TA][4][TB][4];
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < TB; ++b)
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[a][e][b][f] = f32x4{0.f, 0.f, 0.f, 0.f};

        f32x4 ra[D][TA], rb[D][TB];
#pragma unroll
        for (int u = 0; u < D; ++u) {
            if (u < KS) {
#pragma unroll
                for (int a = 0; a < TA; ++a) ra[u][a] = gload<MODE>(pa, offA[a]);
#pragma unroll
                for (int b = 0; b < TB; ++b) rb[u][b] = gload<MODE>(pb, offB[b]);
                pa += sa;
                pb += sb;
            }
        }
        for (int s0 = 0; s0 < KS; s0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int s = s0 + u;
                if (s < KS) {
                    const int younger = min(D - 1, KS - 1 - s);  // k-steps issued after step s and still allowed in flight
                    if (younger == D - 1) wait_vm<LPS*(D - 1)>();
                    else {
                        switch (younger) {  // the last D-1 steps of a tile
                            case 0: wait_vm<0>(); break;
                            case 1: wait_vm<LPS * 1>(); break;
                            case 2: wait_vm<(D > 2 ? LPS * 2 : 0)>(); break;
                            case 3: wait_vm<(D > 3 ? LPS * 3 : 0)>(); break;
                            case 4: wait_vm<(D > 4 ? LPS * 4 : 0)>(); break;
                            case 5: wait_vm<(D > 5 ? LPS * 5 : 0)>(); break;
                            case 6: wait_vm<(D > 6 ? LPS * 6 : 0)>(); break;
                            default: wait_vm<(D > 7 ? LPS * 7 : 0)>(); break;
                        }
                    }
#pragma unroll
                    for (int a = 0; a < TA; ++a) asm volatile("" : "+v"(ra[u][a]));
#pragma unroll
                    for (int b = 0; b < TB; ++b) asm volatile("" : "+v"(rb[u][b]));
#pragma unroll
                    for (int a = 0; a < TA; ++a)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int b = 0; b < TB; ++b)
#pragma unroll
                                for (int f = 0; f < 4; ++f)
                                    acc[a][e][b][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u][a][e], rb[u][b][f], acc[a][e][b][f], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + D < KS) {
#pragma unroll
                        for (int a = 0; a < TA; ++a) ra[u][a] = gload<MODE>(pa, offA[a]);
#pragma unroll
                        for (int b = 0; b < TB; ++b) rb[u][b] = gload<MODE>(pb, offB[b]);
                        pa += sa;
                        pb += sb;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // epilogue: acc[a][e][b][f][t] = C[m0 + 64a + 16q + 4t + e][n0 + 64b + 4r + f]: 16 bytes per lane along N
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + 64 * a + 16 * q + 4 * t + e;
#pragma unroll
                    for (int b = 0; b < TB; ++b) {
                        const int n = n0 + 64 * b + 4 * r;
                        const f32x4 v = {acc[a][e][b][0][t], acc[a][e][b][1][t], acc[a][e][b][2][t], acc[a][e][b][3][t]};
                        if (m < g.M && (!(MODE & 2) || v.x == 1.2345f)) {
                            float* p = g.C + (int64_t)m * g.ldc + n;
                            if (n + 3 < g.N) *reinterpret_cast<f32x4_u*>(p) = v;
                            else
                                for (int k = 0; k < 4; ++k)
                                    if (n + k < g.N) p[k] = v[k];
                        }
                    }
                }
        tile = __builtin_amdgcn_readfirstlane(tn_);
    }
    // every wave makes exactly one dequeue that fails: the one that draws the last ticket resets the counter
    if (lane == 0 && (unsigned)tile == (unsigned)(ntiles + g.n_waves - 1)) atomicExch(ctr, 0u);
}

template <int TA, int TB, int D, int WPS, int MODE = 0>
float run(DrArgs g, int reps, const char* name) {
    g.tiles_m = (g.M + 64 * TA - 1) / (64 * TA);
    g.tiles_n = (g.N + 64 * TB - 1) / (64 * TB);
    g.m_fastest = g.tiles_m <= g.tiles_n;
    const int blocks = 256 * WPS;
    g.n_waves = blocks * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((dr_tn_kernel<TA, TB, D, WPS, MODE>), dim3(blocks), dim3(256), 0, 0, g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((dr_tn_kernel<TA, TB, D, WPS, MODE>), dim3(blocks), dim3(256), 0, 0, g);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-10s mode %d TA=%d TB=%d D=%d waves/SIMD=%d  %.4f ms  %.1f TF  (%d tiles)\n", name, MODE, TA, TB, D, WPS, ms,
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
    BOTH(1, 1, 8, 2);
    BOTH(2, 1, 4, 2);
#define ABL(TA, TB, D, W, MODE) do { run<TA, TB, D, W, MODE>(g2, reps, "dW2"); run<TA, TB, D, W, MODE>(g1, reps, "dW1"); } while (0)
    ABL(1, 1, 8, 2, 1);
    ABL(1, 1, 8, 2, 2);
    ABL(1, 1, 8, 2, 3);
    ABL(1, 1, 8, 2, 4);
    ABL(1, 1, 8, 2, 6);
    ABL(1, 1, 8, 1, 6);
    ABL(2, 1, 4, 2, 1);
    ABL(2, 1, 4, 2, 3);
    ABL(2, 1, 4, 2, 6);
    ABL(2, 1, 4, 1, 0);
    ABL(2, 1, 6, 1, 0);
    return 0;
}
