// What costs the SpMM its gather rate?  The bare gather loop of tools/gather_probe.hip (32 TB/s from an L2-resident table)
// with the SpMM's ingredients added one at a time: (1) the row index distributed with ds_bpermute from a "batch" register,
// (2) + the weight (second bpermute) and the multiply-add, (3) + a 256-byte result row stored every 3 gather groups,
// (4) + the batch itself loaded from memory (coalesced 8-byte entries, contiguous run per wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ X, unsigned mask, int iters, float* __restrict__ out,
                                             const int2* __restrict__ cw, float* __restrict__ Y, unsigned ymask) {
    const int lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned st = wave * 2654435761u + lane * 40503u + 12345u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int2* run = cw + (size_t)wave * iters * 16 + lane;  // MODE 4: 64 entries per batch, 4 groups (16 steps) per batch
    int2 cur = {0, 0};
    if (MODE >= 4) cur = run[0];
    for (int it = 0; it < iters; ++it) {
        const int q = it & 3;
        int c_b, w_b;
        if (MODE >= 4) {
            c_b = cur.x & (int)mask;
            w_b = cur.y;
        } else {
            st = st * 1664525u + 1013904223u;  // a fresh "batch" entry per lane
            c_b = (int)((st >> 7) & mask);
            w_b = __float_as_int(1.0f);
        }
        f32x4 x[4];
        float wt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int src = (q * 4 + u) * 4 + g;
            int c;
            if (MODE >= 1) c = __shfl(c_b, src);
            else { st = st * 1664525u + 1013904223u; c = (int)((st >> 7) & mask); }
            wt[u] = MODE >= 2 ? __int_as_float(__shfl(w_b, src)) : 1.0f;
            x[u] = *reinterpret_cast<const f32x4*>(X + (size_t)(unsigned)c * 64 + gl * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += wt[u] * x[u];
        if (MODE >= 3 && (it % 3) == 2) {
            st = st * 1664525u + 1013904223u;
            const unsigned r = __shfl((st >> 7) & ymask, g * 16);
            *reinterpret_cast<f32x4*>(Y + (size_t)r * 64 + gl * 4) = acc;
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (MODE >= 4 && q == 3) cur = run[(size_t)((it + 1) >> 2) * 64];
    }
    out[(size_t)(blockIdx.x * 256 + threadIdx.x)] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
float run(const float* X, unsigned rows, int blocks, int iters, float* out, const int2* cw, float* Y, unsigned yrows) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(256), 0, 0, X, rows - 1, iters, out, cw, Y, yrows - 1);
    hipEventRecord(a, 0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(256), 0, 0, X, rows - 1, iters, out, cw, Y, yrows - 1);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    float *X, *out, *Y;
    int2* cw;
    const size_t xrows = 1u << 20, yrows = 1u << 20;
    hipMalloc(&X, xrows * 256);
    hipMemset(X, 0, xrows * 256);
    hipMalloc(&Y, yrows * 256);
    hipMalloc(&out, 8192 * 256 * 4);
    const int wpc = 16, blocks = 256 * wpc / 4, waves = blocks * 4;
    const int iters = 128;  // 128 groups x 4 gathers per wave = 512 nonzeros, like a Yelp-shape wave
    hipMalloc(&cw, (size_t)waves * (iters * 16 + 64) * 8);
    {
        size_t n = (size_t)waves * (iters * 16 + 64);
        int2* h = (int2*)malloc(n * 8);
        unsigned s = 1;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i].x = (int)(s >> 7); h[i].y = 0x3f800000; }
        hipMemcpy(cw, h, n * 8, hipMemcpyHostToDevice);
        free(h);
    }
    printf("%-60s %8s %8s\n", "variant (4096 waves x 512 gathers of 256 B, table 2048 rows)", "us", "TB/s");
    const char* names[] = {"0 bare: index from registers", "1 + index through ds_bpermute", "2 + weight through ds_bpermute, multiply-add",
                           "3 + a result row stored every 3 groups (1 Mi-row Y)", "4 + (col, val) batches loaded from a contiguous run"};
    for (unsigned rows : {2048u, 65536u}) {
        for (int mode = 0; mode < 5; ++mode) {
            float ms = 0;
            if (mode == 0) ms = run<0>(X, rows, blocks, iters, out, cw, Y, yrows);
            if (mode == 1) ms = run<1>(X, rows, blocks, iters, out, cw, Y, yrows);
            if (mode == 2) ms = run<2>(X, rows, blocks, iters, out, cw, Y, yrows);
            if (mode == 3) ms = run<3>(X, rows, blocks, iters, out, cw, Y, yrows);
            if (mode == 4) ms = run<4>(X, rows, blocks, iters, out, cw, Y, yrows);
            const double bytes = (double)iters * 4 * waves * 1024.0;
            printf("rows %6u  %-48s %8.1f %8.2f\n", rows, names[mode], ms * 1e3, bytes / ms / 1e9);
        }
    }
    // small Y (L2 resident) for comparison with mode 3/4
    float ms = run<3>(X, 2048, blocks, iters, out, cw, Y, 2048);
    printf("rows   2048  %-48s %8.1f %8.2f\n", "3 with Y confined to 2048 rows", ms * 1e3, (double)iters * 4 * waves * 1024.0 / ms / 1e9);
    return 0;
}
