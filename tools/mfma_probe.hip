// Sustained matrix-pipe rate of MI355X with operands in registers (no LDS, no global memory in the loop): every wave runs
// ITER x 16 back-to-back MFMAs on 16 independent accumulators, operands random.  One or two waves per SIMD, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/bin/mfma_probe && tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 65535]; b[i] = in[(t * 8 + 4 + i) & 65535]; }
    if (KIND == 0) {  // v_mfma_f32_16x16x4_f32
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        f32x4 s = acc[0];
        for (int i = 1; i < 16; ++i) s += acc[i];
        out[t] = s.x + s.y + s.z + s.w;
    } else if (KIND == 1) {  // v_mfma_f32_32x32x2_f32
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 3], b[(i >> 1) & 3], acc[i & 3], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        out[t] = s;
    } else {  // v_mfma_f32_16x16x32_bf16
        bf16x8 xa[4], xb[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { xa[i][e] = (__bf16)(a[i] * (e + 1)); xb[i][e] = (__bf16)(b[i] - e); }
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i & 3], xb[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        f32x4 s = acc[0];
        for (int i = 1; i < 16; ++i) s += acc[i];
        out[t] = s.x + s.y + s.z + s.w;
    }
}

template <int KIND>
void run(const char* name, double flop_per_mfma, int per_iter, const float* in, float* out) {
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        for (int iters : {2000, 20000}) {
            const int blocks = 256 * wgs_per_cu;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
            hipEventRecord(e0, 0);
            const int reps = 5;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= reps;
            const double flops = (double)blocks * 4 * iters * per_iter * flop_per_mfma;
            printf("%-26s waves/SIMD %d  iters %6d  %8.3f ms  %8.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name,
                   wgs_per_cu, iters, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * per_iter * wgs_per_cu));
        }
    }
}

int main() {
    float *in, *out;
    hipMalloc(&in, 65536 * 4);
    hipMalloc(&out, 512 * 256 * 4);
    float* h = (float*)malloc(65536 * 4);
    unsigned st = 12345;
    for (int i = 0; i < 65536; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) & 0xffff) / 65536.f - 0.5f; }
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    run<0>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4, 16, in, out);
    run<1>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, 8, in, out);
    run<2>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, 16, in, out);
    return 0;
}
