// Raw issue-rate ceiling of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 in the shape the GEMM uses.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 r = acc[0];
    for (int i = 1; i < NACC; ++i) r += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y + r.z + r.w;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) r += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
// the GEMM kernel's pattern: TM x TN accumulators, distinct A/B operand registers per k-step, optional fences
template <int TM, int TN, bool FENCE, bool VARY>
__global__ __launch_bounds__(256) void kpat(float* out, int iters, float a0, float b0) {
    f32x4 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    float fa[2][TM][4], fb[2][TN][4];
    for (int p = 0; p < 2; ++p) {
        for (int i = 0; i < TM; ++i) for (int s = 0; s < 4; ++s) fa[p][i][s] = a0 + threadIdx.x + i + s + p;
        for (int j = 0; j < TN; ++j) for (int s = 0; s < 4; ++s) fb[p][j][s] = b0 + j - s + p;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int gi = 0; gi < 8; ++gi) {
            const int par = VARY ? ((gi >> 2) & 1) : 0, sg = VARY ? (gi & 3) : 0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[par][VARY ? i : 0][sg], fb[par][VARY ? j : 0][sg], acc[i][j], 0, 0, 0);
            if (FENCE) __builtin_amdgcn_sched_barrier(0);
        }
        if (VARY) {  // keep the operands opaque like freshly loaded fragments
            for (int i = 0; i < TM; ++i) for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(fa[0][i][s]), "+v"(fa[1][i][s]));
            for (int j = 0; j < TN; ++j) for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(fb[0][j][s]), "+v"(fb[1][j][s]));
        }
    }
    f32x4 r = {0, 0, 0, 0};
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) r += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y + r.z + r.w;
}
template <class K>
void run(const char* name, K kern, int blocks, int iters, double flop_per_mfma, int mfma_per_iter, size_t lds) {
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, iters, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * mfma_per_iter * flop_per_mfma;
    printf("%-34s blocks %4d  %.3f ms  %.1f TF\n", name, blocks, ms, fl / ms / 1e9);
    hipFree(out);
}
int main() {
    const int it = 2000;
    hipFuncSetAttribute((const void*)k16<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    run("16x16x4 10acc 1 WG/CU (1 wave/SIMD)", k16<10>, 256, it, 2048, 80, 100 * 1024);
    run("16x16x4 10acc 2 WG/CU (2 waves/SIMD)", k16<10>, 512, it, 2048, 80, 60 * 1024);
    run("16x16x4 10acc 4 WG/CU", k16<10>, 1024, it, 2048, 80, 0);
    run("16x16x4 16acc 2 WG/CU", k16<16>, 512, it, 2048, 128, 60 * 1024);
    run("16x16x4 4acc  2 WG/CU", k16<4>, 512, it, 2048, 32, 60 * 1024);
    run("16x16x4 2acc  1 WG/CU", k16<2>, 256, it, 2048, 16, 100 * 1024);
    run("32x32x2 4acc  1 WG/CU", k32<4>, 256, it, 4096, 32, 100 * 1024);
    run("32x32x2 4acc  2 WG/CU", k32<4>, 512, it, 4096, 32, 60 * 1024);
    run("pattern 5x2 same operands, no fence 2WG", kpat<5, 2, false, false>, 512, it, 2048, 80, 60 * 1024);
    run("pattern 5x2 varied operands, no fence 2WG", kpat<5, 2, false, true>, 512, it, 2048, 80, 60 * 1024);
    run("pattern 5x2 varied operands, fences  2WG", kpat<5, 2, true, true>, 512, it, 2048, 80, 60 * 1024);
    run("pattern 5x2 varied operands, fences  1WG", kpat<5, 2, true, true>, 256, it, 2048, 80, 100 * 1024);
    run("pattern 4x4 varied operands, fences  2WG", kpat<4, 4, true, true>, 512, it, 2048, 128, 60 * 1024);
    return 0;
}
