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
#include "common.h"

namespace {

constexpr int NTHREADS = 256;

template <int LAY, int R, int BK>
struct TileGeom {
    static constexpr int LD = (LAY == GD_LAY_KC) ? (BK + 4) : (R + ((48 - (R % 32)) % 32));
    static constexpr int FLOATS = (LAY == GD_LAY_KC) ? R * LD : BK * LD;
    static constexpr int F4 = R * BK / 4;
    static constexpr int NL = (F4 + NTHREADS - 1) / NTHREADS;
};

// ---- global -> register staging -----------------------------------------------------------
// 16-byte vectors that are only 4-byte aligned: gfx950 runs with unaligned access enabled and
// hipcc emits global_load_dwordx4 for them, so odd row strides (nn.Linear in_features = 34405)
// still stream with the widest load.
typedef f32x4 f32x4_u __attribute__((aligned(4)));

template <int LAY, int R, int BK>
struct TileStage {
    using G = TileGeom<LAY, R, BK>;
    f32x4 reg[G::NL];

    // Interior K tile (k0 + BK <= kend).  Branch-free so that all loads issue back to back and the
    // only wait is at the LDS write after the MFMAs.  Rows past `nrows` are clamped onto valid rows:
    // their products land in accumulator rows/columns that the epilogue never stores.
    // MC layout needs nrows >= 4 (checked by the caller).
    __device__ __forceinline__ void load_fast(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0,
                                              bool rows_full, int tid) {
#pragma unroll
        for (int i = 0; i < G::NL; ++i) {
            int idx = tid + i * NTHREADS;
            if (G::NL * NTHREADS != G::F4) idx = min(idx, G::F4 - 1);
            if (LAY == GD_LAY_KC) {
                const int r = idx / (BK / 4), kk = (idx % (BK / 4)) * 4;
                const int gr = min(row0 + r, nrows - 1);
                reg[i] = *reinterpret_cast<const f32x4_u*>(P + (int64_t)gr * ld + (k0 + kk));
            } else {
                const int kk = idx / (R / 4), r = (idx % (R / 4)) * 4;
                const int gr = row0 + r;
                const int gc = min(gr, nrows - 4);
                f32x4 v = *reinterpret_cast<const f32x4_u*>(P + (int64_t)(k0 + kk) * ld + gc);
                if (!rows_full) {  // wave-uniform; a vector straddling the last column is shifted in registers
                    const int sh = gr - gc;
                    v.x = sh == 0 ? v.x : (sh == 1 ? v.y : (sh == 2 ? v.z : v.w));
                    v.y = sh == 0 ? v.y : (sh == 1 ? v.z : v.w);
                    v.z = sh == 0 ? v.z : v.w;
                }
                reg[i] = v;
            }
        }
    }

    // Partial K tile (or tiny matrices): element-wise predicated, zero filled.
    __device__ __forceinline__ void load_slow(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0,
                                              int kend, int tid) {
#pragma unroll
        for (int i = 0; i < G::NL; ++i) {
            const int idx = tid + i * NTHREADS;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (G::NL * NTHREADS == G::F4 || idx < G::F4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int gr, gk;
                    if (LAY == GD_LAY_KC) {
                        gr = row0 + idx / (BK / 4);
                        gk = k0 + (idx % (BK / 4)) * 4 + j;
                    } else {
                        gk = k0 + idx / (R / 4);
                        gr = row0 + (idx % (R / 4)) * 4 + j;
                    }
                    if (gr < nrows && gk < kend)
                        v[j] = (LAY == GD_LAY_KC) ? P[(int64_t)gr * ld + gk] : P[(int64_t)gk * ld + gr];
                }
            }
            reg[i] = v;
        }
    }

    __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0,
                                         int kend, bool rows_full, int tid) {
        const bool fast = (k0 + BK <= kend) && (LAY == GD_LAY_KC || nrows >= 4);
        if (fast)
            load_fast(P, ld, row0, nrows, k0, rows_full, tid);
        else
            load_slow(P, ld, row0, nrows, k0, kend, tid);
    }

    __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
#pragma unroll
        for (int i = 0; i < G::NL; ++i) {
            const int idx = tid + i * NTHREADS;
            if (G::NL * NTHREADS == G::F4 || idx < G::F4) {
                if (LAY == GD_LAY_KC) {
                    const int r = idx / (BK / 4), kk = (idx % (BK / 4)) * 4;
                    *reinterpret_cast<f32x4*>(&lds[r * G::LD + kk]) = reg[i];
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
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (LAY == GD_LAY_KC) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[(row_base + t * 16 + r) * G::LD + 16 * c + 4 * q]);
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

__device__ __forceinline__ float gd_tanh(float x) { return tanhf(x); }

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

    TileStage<LAYA, BM, BK> sa;
    TileStage<LAYB, BN, BK> sb;

    const bool a_full = (m0 + BM <= g.M), b_full = (n0 + BN <= g.N);
    if (nt > 0) {
        sa.load(g.A, g.lda, m0, g.M, kbeg, kend, a_full, tid);
        sb.load(g.B, g.ldb, n0, g.N, kbeg, kend, b_full, tid);
        sa.store(smem, tid);
        sb.store(smem + GA::FLOATS, tid);
    }
    __syncthreads();

    for (int it = 0; it < nt; ++it) {
        const float* As = smem + (it & 1) * STAGE_FLOATS;
        const float* Bs = As + GA::FLOATS;
        const bool more = (it + 1 < nt);
        if (more) {
            const int k0 = kbeg + (it + 1) * BK;
            sa.load(g.A, g.lda, m0, g.M, k0, kend, a_full, tid);
            sb.load(g.B, g.ldb, n0, g.N, k0, kend, b_full, tid);
        }
#pragma unroll
        for (int c = 0; c < BK / 16; ++c) {
            float fa[TM][4], fb[TN][4];
            load_frag<LAYA, BM, BK, TM>(As, wm0, c, r, q, fa);
            load_frag<LAYB, BN, BK, TN>(Bs, wn0, c, r, q, fb);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        if (more) {
            float* An = smem + ((it + 1) & 1) * STAGE_FLOATS;
            sa.store(An, tid);
            sb.store(An + GA::FLOATS, tid);
        }
        __syncthreads();
    }

    // ---- epilogue.  acc[i][j][e] = C[m0+wm0+16i+4q+e][n0+wn0+16j+r] ----
    float rowacc[TM][4];
    if (EPI == GD_EPI_LOSS) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) rowacc[i][e] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + wm0 + 16 * i + 4 * q + e;
            if (m < g.M) {
                float c1 = 0.f, c2 = 0.f, p1 = 0.f, p2 = 0.f, sg = 0.f, alpha = 1.f;
                if (EPI == GD_EPI_POST) {
                    c1 = g.r0[m];
                    c2 = g.r1[m];
                    if (g.r2) {
                        p1 = g.r2[m];
                        p2 = g.r3[m];
                    }
                    if (g.aux2) sg = g.r4[m];
                }
                if (EPI == GD_EPI_LOSS && g.r0) alpha = g.r0[m];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn0 + 16 * j + r;
                    if (n < g.N) {
                        float v = acc[i][j][e];
                        if (EPI == GD_EPI_SLAB) {
                            g.C[(int64_t)split * g.slab_stride + (int64_t)m * g.ldc + n] = v;
                        } else if (EPI == GD_EPI_STORE) {
                            float* p = &g.C[(int64_t)m * g.ldc + n];
                            *p = g.accumulate ? (*p + v) : v;
                        } else if (EPI == GD_EPI_BIAS_ACT) {
                            if (g.bias) v += g.bias[n];
                            if (g.act == 1) v = gd_tanh(v);
                            g.C[(int64_t)m * g.ldc + n] = v;
                        } else if (EPI == GD_EPI_LOSS) {
                            if (g.bias) v += g.bias[n];
                            if (g.out2) g.out2[(int64_t)m * g.ldout2 + n] = v;
                            const float d = alpha * v - g.aux[(int64_t)m * g.ldaux + n];
                            g.C[(int64_t)m * g.ldc + n] = d;
                            rowacc[i][e] += d * d;
                        } else if (EPI == GD_EPI_POST) {
                            if (g.bias) v += g.bias[n];
                            const float xt = g.aux[(int64_t)m * g.ldaux + n];
                            float pred = v;
                            if (g.r2) pred = p1 * xt - p2 * v;
                            if (g.out2) g.out2[(int64_t)m * g.ldout2 + n] = pred;
                            float mean = c1 * pred + c2 * xt;
                            if (g.aux2) mean += sg * g.aux2[(int64_t)m * g.ldaux2 + n];
                            g.C[(int64_t)m * g.ldc + n] = mean;
                        }
                    }
                }
            }
        }
    }
    if (EPI == GD_EPI_LOSS) {
        // per-row sum of squares: 16 lanes (r) of each q-group hold one row's columns
        float* rs = smem;  // [BM][WAVES_N]; the tile buffers are dead after the final barrier
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = rowacc[i][e];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                if (r == 0) rs[(wm0 + 16 * i + 4 * q + e) * WAVES_N + (wave % WAVES_N)] = v;
            }
        __syncthreads();
        if (tid < BM && m0 + tid < g.M) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_N; ++w) s += rs[tid * WAVES_N + w];
            g.rowpart[(int64_t)(m0 + tid) * g.ld_rowpart + tile_n] = s;
        }
    }
}

template <int LAYA, int LAYB, int BM, int BN, int BK, int WM, int WN, int EPI>
int launch_one(GdGemm& g, hipStream_t s) {
    using GA = TileGeom<LAYA, BM, BK>;
    using GB = TileGeom<LAYB, BN, BK>;
    constexpr size_t lds = (size_t)2 * (GA::FLOATS + GB::FLOATS) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = gemm_f32_kernel<LAYA, LAYB, BM, BN, BK, WM, WN, EPI>;
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), lds, s, g);
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

int gd_gemm_tile_m(int cls) { return cls == 0 ? 80 : (cls == 1 ? 128 : 64); }
int gd_gemm_tile_n(int cls) { return cls == 2 ? 64 : 128; }
int gd_gemm_bk(int layA, int layB) { return (layA == GD_LAY_MC && layB == GD_LAY_MC) ? 16 : 32; }

int gd_pick_shape_class(int M, int N) {
    if (M <= 64 || N <= 64) return 2;
    const int pad80 = gd_cdiv(M, 80) * 80, pad128 = gd_cdiv(M, 128) * 128;
    return pad80 < pad128 ? 0 : 1;
}

int gd_gemm_launch(int layA, int layB, int epi, int cls, GdGemm& g, hipStream_t s) {
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
        if (epi == GD_EPI_STORE) return launch_class<GD_LAY_MC, GD_LAY_MC, 16, GD_EPI_STORE>(cls, g, s);
    }
    gdmcf_set_error("unsupported gemm variant (layA=%d layB=%d epi=%d)", layA, layB, epi);
    return GDMCF_E_UNSUPPORTED;
}
