// Raw ceiling of 256-byte row gathers on MI355X: no CSR, no shuffles -- every 16-lane group draws a random row of a
// table (LCG in registers), UN independent gathers in flight per wave, results summed.  Varies the table size (where
// the rows are served from: L1 / L2 / Infinity Cache), the waves per CU and the gathers in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UN, bool NT>
__global__ __launch_bounds__(256) void gather_probe(const float* __restrict__ X, unsigned mask, int iters, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15;
    unsigned st = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + g * 40503u + 12345u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f32x4 x[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            st = st * 1664525u + 1013904223u;
            const unsigned r = (st >> 7) & mask;
            const f32x4* p = reinterpret_cast<const f32x4*>(X + (size_t)r * 64 + gl * 4);
            x[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += x[u];
    }
    out[(size_t)(blockIdx.x * 256 + threadIdx.x)] = acc.x + acc.y + acc.z + acc.w;
}

template <int UN, bool NT>
float run(const float* X, unsigned rows, int blocks, int iters, float* out) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((gather_probe<UN, NT>), dim3(blocks), dim3(256), 0, 0, X, rows - 1, iters, out);
    hipEventRecord(a, 0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((gather_probe<UN, NT>), dim3(blocks), dim3(256), 0, 0, X, rows - 1, iters, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const size_t max_rows = 1u << 22;  // 1 GiB
    float *X, *out;
    hipMalloc(&X, max_rows * 256);
    hipMemset(X, 0, max_rows * 256);
    hipMalloc(&out, 4096 * 256 * 4 * 8);
    printf("%10s %8s %6s %4s %4s %10s %10s\n", "table", "rows", "waves", "UN", "nt", "us", "TB/s");
    const unsigned tables[] = {64, 2048, 8192, 16384, 65536, 131072, 1u << 20, 1u << 22};
    for (unsigned rows : tables) {
        for (int wpc : {8, 16, 32}) {
            const int blocks = 256 * wpc / 4;
            for (int un : {4, 8, 16}) {
                for (int nt = 0; nt < 2; ++nt) {
                    if (nt && (wpc != 16)) continue;
                    const long total_instr = 1L << 21;  // wave-instructions of 1 KiB -> 2 GiB gathered
                    const int iters = (int)(total_instr / ((long)blocks * 4 * un));
                    float ms = 0;
                    if (un == 4) ms = nt ? run<4, true>(X, rows, blocks, iters, out) : run<4, false>(X, rows, blocks, iters, out);
                    if (un == 8) ms = nt ? run<8, true>(X, rows, blocks, iters, out) : run<8, false>(X, rows, blocks, iters, out);
                    if (un == 16) ms = nt ? run<16, true>(X, rows, blocks, iters, out) : run<16, false>(X, rows, blocks, iters, out);
                    const double bytes = (double)iters * un * blocks * 4 * 1024.0;
                    printf("%8.1fMB %8u %6d %4d %4d %10.1f %10.2f\n", rows * 256 / 1e6, rows, wpc, un, nt, ms * 1e3, bytes / ms / 1e9);
                }
            }
        }
    }
    return 0;
}
