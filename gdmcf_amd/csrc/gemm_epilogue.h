// Epilogue shared by every MFMA GEMM kernel of the library (f32 kernels in gemm_f32.hip, bf16-input kernel in
// gemm_bf16.hip).  All of them produce 16x16 accumulator blocks in the same register layout:
//   acc[i][j][e] = C[m0 + wm0 + 16 i + 4 q + e][n0 + wn0 + 16 j + r],   r = lane & 15, q = lane >> 4.
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ float gd_tanh(float x) { return tanhf(x); }

// float -> bfloat16 bits, round to nearest even (bf16 shadow of a result, GdGemm::C16)
__device__ __forceinline__ unsigned short gd_epi_bf16(float x) {
    __bf16 h = (__bf16)x;
    return __builtin_bit_cast(unsigned short, h);
}

// ---- row-contiguous epilogue for the weight-gradient products (used for GD_EPI_ADAMW; handles GD_EPI_STORE too) ----
// The MFMA accumulator layout gives every store instruction four 64-byte row segments (dword per lane).  Here the
// tile is handed through LDS instead (the operand buffers are dead after the final barrier) and written back as
// 16 bytes per lane along the rows: a pass of NTH threads covers NTH/(BN/4) full tile rows of BN*4 contiguous bytes.
// With GD_EPI_ADAMW the same pass streams W, exp_avg and exp_avg_sq exactly like the stand-alone AdamW kernel.
// LDS image [slot][BN + 4]: (4*LD) % 32 == 16, so the four lane groups of a ds_write_b32 hit disjoint banks.
// Rounds of IB 16-row blocks per wave row when the whole tile does not fit in LDS_FLOATS.
typedef f32x4 f32x4_ua __attribute__((aligned(4)));

typedef unsigned int gd_u32x2 __attribute__((ext_vector_type(2)));
typedef gd_u32x2 gd_u32x2_ua __attribute__((aligned(2)));

template <int BM, int BN, int TM, int TN, int WAVES_M, int WAVES_N, int EPI, int LDS_FLOATS, int NTH>
__device__ __forceinline__ void gemm_epilogue_rows(f32x4 (&acc)[TM][TN], const GdGemm& g, int m0, int n0, int wn0,
                                                   int r, int q, int wave, int tid, float* smem) {
    static_assert(EPI == GD_EPI_STORE || EPI == GD_EPI_ADAMW, "row epilogue: weight-gradient products only");
    GdAdamHyper hy = g.adam;  // (a step replayed from a hipGraph: this step's scalars come from the device)
    if (EPI == GD_EPI_ADAMW && g.adam_dev) hy = *g.adam_dev;
    constexpr int LD = BN + 4;
    constexpr int IBMAX = LDS_FLOATS / (WAVES_M * 16 * LD);
    static_assert(IBMAX >= 1, "row epilogue: LDS too small for one 16-row block per wave row");
    constexpr int IB = IBMAX < TM ? IBMAX : TM;
    constexpr int TPR = BN / 4, RPP = NTH / TPR, SLOTS = WAVES_M * IB * 16;
    static_assert(NTH % TPR == 0 && BN % 4 == 0, "row epilogue: thread mapping");
    const int wr = wave / WAVES_N;
    const int c4 = (tid % TPR) * 4;
    const int n = n0 + c4;
    float* __restrict__ P = g.C;
    float* __restrict__ Mo = const_cast<float*>(g.aux);
    float* __restrict__ Vo = const_cast<float*>(g.aux2);
    // every row of W / exp_avg / exp_avg_sq starts on a 128-byte line (a tile's piece of a row then covers whole lines)
    const bool lines = EPI == GD_EPI_ADAMW && (g.ldc & 31) == 0 && (((uintptr_t)P | (uintptr_t)Mo | (uintptr_t)Vo) & 127) == 0;
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += IB) {
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) {
            if (i0 + ib < TM) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        smem[((wr * IB + ib) * 16 + 4 * q + e) * LD + wn0 + 16 * j + r] = acc[(i0 + ib) < TM ? (i0 + ib) : 0][j][e];
            }
        }
        __syncthreads();
        if (EPI == GD_EPI_ADAMW && n + 3 < g.N) {
            // Fused optimiser, full 16-byte groups: U row slots per trip with ALL their parameter / moment loads issued before the
            // first one is consumed.  One slot per trip left 48 bytes per thread in flight -- the epilogue then streams its
            // 26-28 bytes per parameter at ~4 TB/s (rocprofv3: 0.63 ms per Amazon-Book weight in bf16 mode, 3.9 TB/s) where the
            // stand-alone AdamW kernel reaches 6.3; loads use row indices clamped into the matrix, only the stores are predicated.
            constexpr int U = 4;  // (6 and 8 slots per trip measured slower: 0.522 / 0.542 against 0.509 ms per Amazon-Book weight)
            for (int s0 = tid / TPR; s0 < SLOTS; s0 += RPP * U) {
                f32x4 pv[U], mv[U], vv[U], gq[U];
                int64_t oo[U];
                int mm[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int sl = min(s0 + u * RPP, SLOTS - 1);
                    const int swr = sl / (IB * 16), sib = (sl >> 4) % IB;
                    const int m = m0 + swr * (BM / WAVES_M) + 16 * (i0 + sib) + (sl & 15);
                    ok[u] = (s0 + u * RPP < SLOTS) && (i0 + sib < TM) && (m < g.M);
                    mm[u] = min(m, g.M - 1);
                    oo[u] = (int64_t)mm[u] * g.ldc + n;
                    // (rows that start anywhere: plain accesses -- nontemporal ones measured 2-10 % slower, a piece's first and last
                    // line are shared with the neighbouring tiles.  Rows on 128-byte lines, as FusedAdamW.fuse_into_backward seats
                    // them: nontemporal loads and stores, configs[2] 1.79 -> 1.70 ms per step, round 4)
                    if (lines) {
                        pv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_ua*>(P + oo[u]));
                        mv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_ua*>(Mo + oo[u]));
                        vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_ua*>(Vo + oo[u]));
                    } else {
                        pv[u] = *reinterpret_cast<const f32x4_ua*>(P + oo[u]);
                        mv[u] = *reinterpret_cast<const f32x4_ua*>(Mo + oo[u]);
                        vv[u] = *reinterpret_cast<const f32x4_ua*>(Vo + oo[u]);
                    }
                    gq[u] = *reinterpret_cast<const f32x4*>(&smem[sl * LD + c4]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!ok[u]) continue;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float pk = pv[u][k], mk = mv[u][k], vk = vv[u][k];
                        gd_adam_elem(pk, gq[u][k], mk, vk, hy);
                        pv[u][k] = pk;
                        mv[u][k] = mk;
                        vv[u][k] = vk;
                    }
                    if (lines) {
                        __builtin_nontemporal_store(pv[u], reinterpret_cast<f32x4_ua*>(P + oo[u]));
                        __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4_ua*>(Mo + oo[u]));
                        __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4_ua*>(Vo + oo[u]));
                    } else {
                        *reinterpret_cast<f32x4_ua*>(P + oo[u]) = pv[u];
                        *reinterpret_cast<f32x4_ua*>(Mo + oo[u]) = mv[u];
                        *reinterpret_cast<f32x4_ua*>(Vo + oo[u]) = vv[u];
                    }
                    if (g.C16) {
                        const gd_u32x2 w16 = {gd_epi_bf16(pv[u][0]) | ((unsigned)gd_epi_bf16(pv[u][1]) << 16),
                                              gd_epi_bf16(pv[u][2]) | ((unsigned)gd_epi_bf16(pv[u][3]) << 16)};
                        *reinterpret_cast<gd_u32x2_ua*>(static_cast<unsigned short*>(g.C16) + (int64_t)mm[u] * g.ldc16 + n) = w16;
                    }
                }
            }
        } else  // (the common barrier below: the two paths may split a wave in the last column tile)
        for (int s = tid / TPR; s < SLOTS; s += RPP) {
            const int swr = s / (IB * 16), sib = (s >> 4) % IB;
            const int m = m0 + swr * (BM / WAVES_M) + 16 * (i0 + sib) + (s & 15);
            if (i0 + sib >= TM || m >= g.M || n >= g.N) continue;
            const f32x4 gv = *reinterpret_cast<const f32x4*>(&smem[s * LD + c4]);
            const int64_t o = (int64_t)m * g.ldc + n;
            if (n + 3 < g.N) {
                if (EPI == GD_EPI_STORE) {
                    f32x4 v = gv;
                    if (g.accumulate) v += *reinterpret_cast<const f32x4_ua*>(P + o);
                    *reinterpret_cast<f32x4_ua*>(P + o) = v;
                } else {
                    f32x4 pv = *reinterpret_cast<const f32x4_ua*>(P + o);
                    f32x4 mv = *reinterpret_cast<const f32x4_ua*>(Mo + o);
                    f32x4 vv = *reinterpret_cast<const f32x4_ua*>(Vo + o);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float pk = pv[k], mk = mv[k], vk = vv[k];
                        gd_adam_elem(pk, gv[k], mk, vk, hy);
                        pv[k] = pk;
                        mv[k] = mk;
                        vv[k] = vk;
                    }
                    *reinterpret_cast<f32x4_ua*>(P + o) = pv;
                    *reinterpret_cast<f32x4_ua*>(Mo + o) = mv;
                    *reinterpret_cast<f32x4_ua*>(Vo + o) = vv;
                    if (g.C16) {
                        const gd_u32x2 w16 = {gd_epi_bf16(pv[0]) | ((unsigned)gd_epi_bf16(pv[1]) << 16),
                                              gd_epi_bf16(pv[2]) | ((unsigned)gd_epi_bf16(pv[3]) << 16)};
                        *reinterpret_cast<gd_u32x2_ua*>(static_cast<unsigned short*>(g.C16) + (int64_t)m * g.ldc16 + n) = w16;
                    }
                }
            } else {
                for (int k = 0; k < 4 && n + k < g.N; ++k) {
                    if (EPI == GD_EPI_STORE) {
                        P[o + k] = g.accumulate ? P[o + k] + gv[k] : gv[k];
                    } else {
                        float pk = P[o + k], mk = Mo[o + k], vk = Vo[o + k];
                        gd_adam_elem(pk, gv[k], mk, vk, hy);
                        P[o + k] = pk;
                        Mo[o + k] = mk;
                        Vo[o + k] = vk;
                        if (g.C16) static_cast<unsigned short*>(g.C16)[(int64_t)m * g.ldc16 + n + k] = gd_epi_bf16(pk);
                    }
                }
            }
        }
        if (i0 + IB < TM) __syncthreads();
    }
}

// ---- epilogue (shared by the plain and the wave-specialised kernel) ---------------------------------------
template <int BM, int TM, int TN, int WAVES_N, int EPI>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[TM][TN], const GdGemm& g, int m0, int n0, int wm0, int wn0,
                                              int r, int q, int split, int tile_n, int wave, int tid, float* smem) {
    // ---- epilogue.  acc[i][j][e] = C[m0+wm0+16i+4q+e][n0+wn0+16j+r] ----
    // Branch-free on the load side: every read (bias, target, x_t, z, per-row coefficients) uses indices
    // clamped into the matrix and is issued before any of them is consumed, so the ~40 loads per lane are
    // in flight together; only the stores are predicated.
    float rowacc[TM][4];
    int ncl[TN];
    bool nok[TN];
    float biasv[TN];
    GdAdamHyper hy = g.adam;
    if (EPI == GD_EPI_ADAMW && g.adam_dev) hy = *g.adam_dev;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + 16 * j + r;
        nok[j] = n < g.N;
        ncl[j] = min(n, g.N - 1);
        biasv[j] = 0.f;
        if ((EPI == GD_EPI_BIAS_ACT || EPI == GD_EPI_LOSS || EPI == GD_EPI_POST) && g.bias) biasv[j] = g.bias[ncl[j]];
    }
    if (EPI == GD_EPI_ADAMW) {
        // Fused optimiser: the tile of the weight gradient never leaves the accumulators.  Per 16-row block:
        // load p, exp_avg, exp_avg_sq (clamped indices, all in flight), update, predicated stores.
        float* __restrict__ P = g.C;
        float* __restrict__ Mo = const_cast<float*>(g.aux);
        float* __restrict__ Vo = const_cast<float*>(g.aux2);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float pv[4][TN], mv[4][TN], vv[4][TN];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mc = min(m0 + wm0 + 16 * i + 4 * q + e, g.M - 1);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int64_t o = (int64_t)mc * g.ldc + ncl[j];
                    pv[e][j] = P[o];
                    mv[e][j] = Mo[o];
                    vv[e][j] = Vo[o];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm0 + 16 * i + 4 * q + e;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    gd_adam_elem(pv[e][j], acc[i][j][e], mv[e][j], vv[e][j], hy);
                    if (m < g.M && nok[j]) {
                        const int64_t o = (int64_t)m * g.ldc + ncl[j];
                        P[o] = pv[e][j];
                        Mo[o] = mv[e][j];
                        Vo[o] = vv[e][j];
                    }
                }
            }
        }
    } else if (EPI == GD_EPI_SLAB || EPI == GD_EPI_STORE || EPI == GD_EPI_BIAS_ACT) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm0 + 16 * i + 4 * q + e;
                if (m < g.M) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (!nok[j]) continue;
                        float v = acc[i][j][e];
                        if (EPI == GD_EPI_SLAB) {
                            g.C[(int64_t)split * g.slab_stride + (int64_t)m * g.ldc + ncl[j]] = v;
                        } else if (EPI == GD_EPI_STORE) {
                            float* p = &g.C[(int64_t)m * g.ldc + ncl[j]];
                            *p = g.accumulate ? (*p + v) : v;
                        } else {
                            v += biasv[j];
                            if (g.act == 1) v = gd_tanh(v);
                            g.C[(int64_t)m * g.ldc + ncl[j]] = v;
                            if (g.C16) static_cast<unsigned short*>(g.C16)[(int64_t)m * g.ldc16 + ncl[j]] = gd_epi_bf16(v);
                        }
                    }
                }
            }
    } else {
        // LOSS / POST, one 16-row block at a time: phase 1 -- all auxiliary loads of the block (4*TN per
        // array, in flight together); phase 2 -- arithmetic + predicated stores.
        const bool has_z = (EPI == GD_EPI_POST) && (g.aux2 != nullptr);
        const bool has_r = (EPI == GD_EPI_POST) && (g.r2 != nullptr);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float av[4][TN];  // target (LOSS) or x_t (POST)
            float zv[4][TN];  // z noise (POST with sampling noise)
            float c1v[4], c2v[4], p1v[4], p2v[4], sgv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mc = min(m0 + wm0 + 16 * i + 4 * q + e, g.M - 1);
                c1v[e] = 1.f; c2v[e] = 0.f; p1v[e] = 0.f; p2v[e] = 0.f; sgv[e] = 0.f;
                if (EPI == GD_EPI_LOSS) {
                    if (g.r0) c1v[e] = g.r0[mc];  // alpha
                } else {
                    c1v[e] = g.r0[mc];
                    c2v[e] = g.r1[mc];
                    if (has_r) {
                        p1v[e] = g.r2[mc];
                        p2v[e] = g.r3[mc];
                    }
                    if (has_z) sgv[e] = g.r4[mc];
                }
                if (EPI == GD_EPI_LOSS && g.aux_bits) {
                    // {0,1} target rows kept as bitmaps (CSR input path).  The wave's columns start at a multiple of 32 and
                    // lane r owns column 16*j + r of it: columns j and j^1 share a word -> one load per 32 columns and row
                    // (the words are shared by the 16 lanes of the row: a broadcast load)
                    static_assert(EPI != GD_EPI_LOSS || TN % 2 == 0, "bitmap target: a wave's columns must start at a multiple of 32");
                    const int cbase = n0 + wn0;
                    uint32_t wv[(TN + 1) / 2];
#pragma unroll
                    for (int jj = 0; jj < (TN + 1) / 2; ++jj)
                        wv[jj] = g.aux_bits[(int64_t)mc * g.ldbits + min((int64_t)((cbase >> 5) + jj), g.ldbits - 1)];
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        av[e][j] = (float)((wv[j >> 1] >> (16 * (j & 1) + r)) & 1u);  // v_bfe_u32 + v_cvt_f32_u32
                        zv[e][j] = 0.f;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        av[e][j] = g.aux[(int64_t)mc * g.ldaux + ncl[j]];
                        zv[e][j] = has_z ? g.aux2[(int64_t)mc * g.ldaux2 + ncl[j]] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm0 + 16 * i + 4 * q + e;
                const bool mok = m < g.M;
                float racc = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const bool ok = mok && nok[j];
                    const float v = acc[i][j][e] + biasv[j];
                    if (EPI == GD_EPI_LOSS) {
                        const float d = c1v[e] * v - av[e][j];
                        if (ok) {
                            if (g.out2) g.out2[(int64_t)m * g.ldout2 + ncl[j]] = v;
                            g.C[(int64_t)m * g.ldc + ncl[j]] = d;
                            if (g.C16) static_cast<unsigned short*>(g.C16)[(int64_t)m * g.ldc16 + ncl[j]] = gd_epi_bf16(d);
                            racc += d * d;
                        }
                    } else {
                        const float xt = av[e][j];
                        const float pred = has_r ? (p1v[e] * xt - p2v[e] * v) : v;
                        float mean = c1v[e] * pred + c2v[e] * xt;
                        if (has_z) mean += sgv[e] * zv[e][j];
                        if (ok) {
                            if (g.out2) g.out2[(int64_t)m * g.ldout2 + ncl[j]] = pred;
                            g.C[(int64_t)m * g.ldc + ncl[j]] = mean;
                            if (g.C16) static_cast<unsigned short*>(g.C16)[(int64_t)m * g.ldc16 + ncl[j]] = gd_epi_bf16(mean);
                        }
                    }
                }
                rowacc[i][e] = racc;
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

}  // namespace
