// Dense products whose shape the vectorised f32 loaders cannot take (a reduction shorter than 4 for K-contiguous
// operands, fewer than 4 rows for row-contiguous ones: a single item, a hidden width of 2, ...).  One workgroup per
// output row, one thread per output element, a k-ordered fmaf chain straight from global memory, the same epilogue
// semantics as gemm_epilogue.h written per element.  Speed is irrelevant at these sizes; what matters is that every
// shape the reference accepts produces a result (reference models/DNN.py:79-86 has no size restriction).
#include "common.h"

namespace {

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(const GdGemm g, int layA, int layB) {
    __shared__ float red[256];
    const int m = blockIdx.x, tid = threadIdx.x;
    float rowacc = 0.f;
    for (int n = tid; n < g.N; n += 256) {
        float acc = 0.f;
        for (int k = 0; k < g.K; ++k) {
            const float a = layA == GD_LAY_KC ? g.A[(int64_t)m * g.lda + k] : g.A[(int64_t)k * g.lda + m];
            const float b = layB == GD_LAY_KC ? g.B[(int64_t)n * g.ldb + k] : g.B[(int64_t)k * g.ldb + n];
            acc = fmaf(a, b, acc);
        }
        const int64_t o = (int64_t)m * g.ldc + n;
        if (EPI == GD_EPI_SLAB) {
            g.C[o] = acc;  // launched with a single split
        } else if (EPI == GD_EPI_BIAS_ACT) {
            float v = acc + (g.bias ? g.bias[n] : 0.f);
            if (g.act == 1) v = tanhf(v);
            g.C[o] = v;
        } else if (EPI == GD_EPI_LOSS) {
            const float v = acc + (g.bias ? g.bias[n] : 0.f);
            const float tgt = g.aux_bits ? (((g.aux_bits[(int64_t)m * g.ldbits + (n >> 5)] >> (n & 31)) & 1u) ? 1.f : 0.f)
                                         : g.aux[(int64_t)m * g.ldaux + n];
            const float d = (g.r0 ? g.r0[m] : 1.f) * v - tgt;
            if (g.out2) g.out2[(int64_t)m * g.ldout2 + n] = v;
            g.C[o] = d;
            rowacc += d * d;
        } else if (EPI == GD_EPI_POST) {
            const float v = acc + (g.bias ? g.bias[n] : 0.f);
            const float xt = g.aux[(int64_t)m * g.ldaux + n];
            const float pred = g.r2 ? (g.r2[m] * xt - g.r3[m] * v) : v;
            float mean = g.r0[m] * pred + g.r1[m] * xt;
            if (g.aux2) mean += g.r4[m] * g.aux2[(int64_t)m * g.ldaux2 + n];
            if (g.out2) g.out2[(int64_t)m * g.ldout2 + n] = pred;
            g.C[o] = mean;
        } else if (EPI == GD_EPI_STORE) {
            g.C[o] = g.accumulate ? g.C[o] + acc : acc;
        } else {  // GD_EPI_ADAMW: C = parameter, aux = exp_avg, aux2 = exp_avg_sq
            float* Mo = const_cast<float*>(g.aux);
            float* Vo = const_cast<float*>(g.aux2);
            float p = g.C[o], mo = Mo[o], vo = Vo[o];
            gd_adam_elem(p, acc, mo, vo, g.adam_dev ? *g.adam_dev : g.adam);
            g.C[o] = p;
            Mo[o] = mo;
            Vo[o] = vo;
        }
    }
    if (EPI == GD_EPI_LOSS) {  // per-row sum of squares: fixed-order tree over the 256 thread partials
        red[tid] = rowacc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) g.rowpart[(int64_t)m * g.ld_rowpart] = red[0];
    }
}

}  // namespace

// g.splits / g.kchunk / g.tiles_* are set so that the callers' reducers (one slab, one row-partial column) still apply
int gd_gemm_small_launch(int layA, int layB, int epi, GdGemm& g, hipStream_t s) {
    g.splits = 1;
    g.kchunk = g.K > 0 ? g.K : 1;
    g.tiles_m = g.M;
    g.tiles_n = 1;
    const dim3 grid((unsigned)g.M), block(256);
    GdProfScope prof(g.prof_tag, 2.0 * g.M * g.N * g.K, s);
    switch (epi) {
        case GD_EPI_SLAB: hipLaunchKernelGGL(gemm_f32_small_kernel<GD_EPI_SLAB>, grid, block, 0, s, g, layA, layB); break;
        case GD_EPI_BIAS_ACT: hipLaunchKernelGGL(gemm_f32_small_kernel<GD_EPI_BIAS_ACT>, grid, block, 0, s, g, layA, layB); break;
        case GD_EPI_LOSS: hipLaunchKernelGGL(gemm_f32_small_kernel<GD_EPI_LOSS>, grid, block, 0, s, g, layA, layB); break;
        case GD_EPI_POST: hipLaunchKernelGGL(gemm_f32_small_kernel<GD_EPI_POST>, grid, block, 0, s, g, layA, layB); break;
        case GD_EPI_STORE: hipLaunchKernelGGL(gemm_f32_small_kernel<GD_EPI_STORE>, grid, block, 0, s, g, layA, layB); break;
        case GD_EPI_ADAMW: hipLaunchKernelGGL(gemm_f32_small_kernel<GD_EPI_ADAMW>, grid, block, 0, s, g, layA, layB); break;
        default: gdmcf_set_error("gemm_small: bad epilogue %d", epi); return GDMCF_E_ARG;
    }
    return gd_launch_status("gemm_f32_small");
}
