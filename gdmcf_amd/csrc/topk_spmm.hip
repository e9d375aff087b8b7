// Evaluation-side kernels: masked per-row top-k (reference main.py:296-301) and the LightGCN CSR
// SpMM (reference lightGCN.py:184-189).  Both are HBM / cache-bandwidth bound.
#include <stdlib.h>

#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// masked top-k: one 256-thread workgroup per row.
//   1. history mask -> bitmap in LDS (I bits)
//   2. 4-pass 8-bit radix select on order-preserving uint32 keys -> k-th largest key
//   3. collect keys > threshold (any order) + the lowest-index ties == threshold
//   4. bitonic sort of (key, ~index) pairs -> descending score, ascending index on ties
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t order_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

constexpr uint32_t NEG_INF_KEY = 0x007FFFFFu;  // order_key(-inf)

__device__ __forceinline__ uint32_t masked_key(const float* __restrict__ row, const uint32_t* bitmap, int i) {
    if (bitmap[i >> 5] & (1u << (i & 31))) return NEG_INF_KEY;
    return order_key(row[i]);
}

// NT threads per row: 1024 when the row is staged in LDS (the staged row allows one workgroup per CU anyway, and
// the select is a chain of short LDS loops -- 16 waves hide their latency, 4 waves measured 2.4x slower)
template <bool STAGE, int NT>
__global__ __launch_bounds__(NT) void topk_kernel(const float* __restrict__ pred, int64_t ldp, int I,
                                                   const int64_t* __restrict__ indptr,
                                                   const int32_t* __restrict__ indices, int k, int KP,
                                                   int64_t* __restrict__ idx_out, float* __restrict__ val_out, int redo) {
    // redo: second launch behind topk_fast_kernel -- only the rows it marked (first index -1) are selected here
    if (redo && idx_out[(int64_t)blockIdx.x * k] != -1) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(smem_raw);  // [KP]
    uint32_t* hist = reinterpret_cast<uint32_t*>(cand + KP);                     // [256]
    uint32_t* ctl = hist + 256;                                                  // [24]: 4 scalars + one count per wave
    uint32_t* bitmap = ctl + 24;                                                 // [ceil(I/32)]
    const int tid = threadIdx.x;
    const int row_id = blockIdx.x;
    const float* row = pred + (int64_t)row_id * ldp;
    const int nwords = (I + 31) >> 5;
    uint32_t* keys = bitmap + nwords;                                            // [I] when STAGE
    // key of element i: from the LDS copy when the row fits (one global pass, 8 loads in flight per thread),
    // otherwise recomputed from global memory in every pass
    auto KEY = [&](int i) -> uint32_t { return STAGE ? keys[i] : masked_key(row, bitmap, i); };

    for (int w = tid; w < nwords; w += NT) bitmap[w] = 0u;
    for (int j = tid; j < KP; j += NT) cand[j] = 0ull;
    __syncthreads();
    if (indptr) {
        const int64_t beg = indptr[row_id], end = indptr[row_id + 1];
        for (int64_t j = beg + tid; j < end; j += NT) {
            const int c = indices[j];
            if (c >= 0 && c < I) atomicOr(&bitmap[c >> 5], 1u << (c & 31));
        }
    }
    __syncthreads();
    if (STAGE) {
        int i = tid;
        for (; i + 7 * NT < I; i += 8 * NT) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[i + u * NT];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = i + u * NT;
                keys[e] = (bitmap[e >> 5] & (1u << (e & 31))) ? NEG_INF_KEY : order_key(v[u]);
            }
        }
        for (; i < I; i += NT) keys[i] = masked_key(row, bitmap, i);
        __syncthreads();
    }

    // radix select
    uint32_t prefix = 0, pmask = 0;
    int need = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0u;
        // Scores cluster (pass 1 sees ~2 exponent bins), so plain LDS atomics would serialise on one counter:
        // every thread counts the bin of a sampled key privately and only the other bins go through atomics.
        const uint32_t guess = (KEY(I >> 1) >> shift) & 255u;
        uint32_t local = 0;
        __syncthreads();
        for (int i = tid; i < I; i += NT) {
            const uint32_t key = KEY(i);
            if ((key & pmask) == prefix) {
                const uint32_t b = (key >> shift) & 255u;
                if (b == guess) ++local;
                else atomicAdd(&hist[b], 1u);
            }
        }
        if (local) atomicAdd(&hist[guess], local);
        __syncthreads();
        if (tid < 64) {
            // find the bin holding the need-th largest key: wave 0 scans the 256 bins from the top, 4 per lane,
            // with a shuffle prefix sum (a serial walk by one thread costs 256 dependent LDS reads per pass)
            const int lane = tid;
            uint32_t c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = hist[255 - 4 * lane - j];
            const uint32_t sum4 = c[0] + c[1] + c[2] + c[3];
            uint32_t pre = sum4;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(pre, o);
                if (lane >= o) pre += t;
            }
            const unsigned long long hit = __ballot(pre >= (uint32_t)need);
            const int L = hit ? (__ffsll((long long)hit) - 1) : 63;
            if (lane == L) {
                uint32_t cum = pre - sum4;
                int j = 0;
                for (; j < 3; ++j) {
                    if (cum + c[j] >= (uint32_t)need) break;
                    cum += c[j];
                }
                ctl[0] = (uint32_t)(255 - 4 * lane - j);
                ctl[1] = cum;   // elements strictly above the bin (within the prefix)
                ctl[2] = c[j];  // elements in the bin
            }
        }
        __syncthreads();
        prefix |= ctl[0] << shift;
        pmask |= 255u << shift;
        need -= (int)ctl[1];
        __syncthreads();
    }
    const uint32_t thr = prefix;       // k-th largest key
    const int n_eq_total = (int)ctl[2];  // elements equal to thr
    const int n_gt = k - need;         // elements strictly greater
    // need = number of ties to take (>= 1)

    if (tid == 0) ctl[3] = 0u;
    __syncthreads();
    if (n_eq_total == need) {
        // no surplus ties: order of collection is irrelevant
        for (int i = tid; i < I; i += NT) {
            const uint32_t key = KEY(i);
            if (key >= thr) {
                const uint32_t slot = atomicAdd(&ctl[3], 1u);
                cand[slot] = ((unsigned long long)key << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
            }
        }
    } else {
        for (int i = tid; i < I; i += NT) {
            const uint32_t key = KEY(i);
            if (key > thr) {
                const uint32_t slot = atomicAdd(&ctl[3], 1u);
                cand[slot] = ((unsigned long long)key << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
            }
        }
        // ties in index order: chunked ordered scan, stop once `need` were taken
        int taken = 0;
        const int lane = tid & 63, wave = tid >> 6;
        for (int base = 0; base < I && taken < need; base += NT) {
            const int i = base + tid;
            const bool f = (i < I) && (KEY(i) == thr);
            const unsigned long long bal = __ballot(f);
            if (lane == 0) ctl[4 + wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            int before = 0;
            for (int w = 0; w < wave; ++w) before += (int)ctl[4 + w];
            int total = 0;
            for (int w = 0; w < NT / 64; ++w) total += (int)ctl[4 + w];
            const int rank = taken + before + (int)__popcll(bal & ((1ull << lane) - 1ull));
            if (f && rank < need)
                cand[n_gt + rank] = ((unsigned long long)thr << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
            taken += total;
            __syncthreads();
        }
    }
    __syncthreads();

    // bitonic sort, descending
    for (int size = 2; size <= KP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int j = tid; j < KP / 2; j += NT) {
                const int lo = ((j / stride) * stride * 2) + (j % stride);
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = cand[lo], b = cand[hi];
                if ((a < b) == desc) {
                    cand[lo] = b;
                    cand[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int j = tid; j < k; j += NT) {
        const unsigned long long c = cand[j];
        idx_out[(int64_t)row_id * k + j] = (int64_t)(0xFFFFFFFFu - (uint32_t)(c & 0xFFFFFFFFull));
        if (val_out) val_out[(int64_t)row_id * k + j] = key_to_float((uint32_t)(c >> 32));
    }
}

// ---------------------------------------------------------------------------------------------
// The fast path (round 4): no radix passes, no staged row.  A thread keeps the upper halves of its KPT keys of the row in REGISTERS and their
// maximum goes to LDS; the k-th largest of the NT thread maxima, L, is a lower bound of the k-th largest key of the row (k
// threads hold an element >= L), so every element of the answer -- and every tie of its last element -- is among the elements
// >= L: for score rows that is k plus a few.  They are collected with their indices and sorted as (key, ~index) pairs: descending
// score, ascending index among equal scores, the first k are the answer -- the same list as topk_kernel's, bit for bit.
// ~14 KB of LDS and < 64 registers: two 1024-thread workgroups per CU, 400 rows in one round; the 55 MB of scores are read once.
// When more than CAP elements are >= L (degenerate rows: all scores equal, the large ones all in a few threads' columns, k above
// the number of unmasked items) the row is marked -- first index -1 -- and topk_kernel selects it in the launch that follows.
// ---------------------------------------------------------------------------------------------
constexpr int TOPK_FAST_CAP = 1024;
template <int NT, int KPT>
__global__ __launch_bounds__(NT, 8) void topk_fast_kernel(const float* __restrict__ pred, int64_t ldp, int I,
                                                         const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                         int k, int64_t* __restrict__ idx_out, float* __restrict__ val_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(smem_raw);  // [TOPK_FAST_CAP]
    uint32_t* tmax = reinterpret_cast<uint32_t*>(cand + TOPK_FAST_CAP);         // [NT]
    uint32_t* ctl = tmax + NT;                                                   // [4]
    uint32_t* bitmap = ctl + 4;                                                  // [ceil(I/32)]
    const int tid = threadIdx.x;
    const int row_id = blockIdx.x;
    const float* row = pred + (int64_t)row_id * ldp;
    const int nwords = (I + 31) >> 5;
    for (int w = tid; w < nwords; w += NT) bitmap[w] = 0u;
    if (tid == 0) ctl[0] = 0u;
    __syncthreads();
    if (indptr) {
        const int64_t beg = indptr[row_id], end = indptr[row_id + 1];
        for (int64_t j = beg + tid; j < end; j += NT) {
            const int c = indices[j];
            if (c >= 0 && c < I) atomicOr(&bitmap[c >> 5], 1u << (c & 31));
        }
    }
    __syncthreads();
    // the thread's elements tid + u NT: the UPPER HALVES of their keys stay in registers, two per register (the exact key of the
    // few elements whose upper half reaches the bound's is fetched again below: L2); eight loads in flight
    uint32_t kh[(KPT + 1) / 2];
#pragma unroll
    for (int u = 0; u < (KPT + 1) / 2; ++u) kh[u] = 0u;
    uint32_t best = 0u;
#pragma unroll
    for (int u0 = 0; u0 < KPT; u0 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + (u0 + j) * NT;
            v[j] = (u0 + j < KPT && i < I) ? row[i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (u0 + j < KPT) {
                const int i = tid + (u0 + j) * NT;
                // (key 0 -- below every real key -- past the end of the row)
                const uint32_t kk = (i < I) ? ((bitmap[i >> 5] & (1u << (i & 31))) ? NEG_INF_KEY : order_key(v[j])) : 0u;
                kh[(u0 + j) >> 1] |= (kk >> 16) << (16 * ((u0 + j) & 1));
                best = kk > best ? kk : best;
            }
        }
    }
    tmax[tid] = best;
    __syncthreads();
    // bitonic sort of the NT maxima, descending
    for (int size = 2; size <= NT; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (tid < NT / 2) {
                const int lo = ((tid / stride) * stride * 2) + (tid % stride);
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const uint32_t a = tmax[lo], b = tmax[hi];
                if ((a < b) == desc) {
                    tmax[lo] = b;
                    tmax[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    const uint32_t L = tmax[k - 1];
    // collect every element >= L
    const uint32_t Lh = L >> 16;
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int i = tid + u * NT;
        const uint32_t h = (kh[u >> 1] >> (16 * (u & 1))) & 0xFFFFu;
        if (i < I && h >= Lh) {
            const uint32_t kk = masked_key(row, bitmap, i);  // the exact key
            if (kk >= L) {
                const uint32_t slot = atomicAdd(&ctl[0], 1u);
                if (slot < (uint32_t)TOPK_FAST_CAP) cand[slot] = ((unsigned long long)kk << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
            }
        }
    }
    __syncthreads();
    const uint32_t C = ctl[0];
    if (C > (uint32_t)TOPK_FAST_CAP) {  // too many: topk_kernel takes this row in the next launch
        if (tid == 0) idx_out[(int64_t)row_id * k] = -1;
        return;
    }
    // sort the candidates (padded with key 0 up to the next power of two >= C, at least k)
    int CP = 2;
    while (CP < (int)C || CP < k) CP <<= 1;
    for (int j = (int)C + tid; j < CP; j += NT) cand[j] = 0ull;
    __syncthreads();
    for (int size = 2; size <= CP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int j = tid; j < CP / 2; j += NT) {
                const int lo = ((j / stride) * stride * 2) + (j % stride);
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = cand[lo], b = cand[hi];
                if ((a < b) == desc) {
                    cand[lo] = b;
                    cand[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int j = tid; j < k; j += NT) {
        const unsigned long long c = cand[j];
        idx_out[(int64_t)row_id * k + j] = (int64_t)(0xFFFFFFFFu - (uint32_t)(c & 0xFFFFFFFFull));
        if (val_out) val_out[(int64_t)row_id * k + j] = key_to_float((uint32_t)(c >> 32));
    }
}

// ---------------------------------------------------------------------------------------------
// CSR SpMM  Y = (A X + sum_k addend_k) * scale,  one wave per VIRTUAL row.
// Degree skew is the problem (a hub item has 1e4+ neighbours, the median row ~20): the host splits
// every row into virtual rows of <= chunk nonzeros (plan arrays below), so hub rows spread over many
// waves; a row that was split writes float partials that a second tiny kernel adds in slot order
// (deterministic, no atomics).  LPR lanes cover one embedding row with 16-byte loads, so a wave
// gathers 64/LPR neighbour rows per instruction (d = 64: 4 x 256 B) and keeps 4 such instructions
// in flight.  The LightGCN layer mean is fused through `addends`/`scale` in the last layer.
// ---------------------------------------------------------------------------------------------
constexpr int SPMM_MAX_ADD = 8;
struct SpmmAdd {
    const float* p[SPMM_MAX_ADD];
    int n;
    int64_t ld;
    float scale;
};

template <int LPR>
__global__ __launch_bounds__(256) void spmm_vec_kernel(const int64_t* __restrict__ vbeg, const int64_t* __restrict__ vend,
                                                       const int32_t* __restrict__ vrow,
                                                       const int32_t* __restrict__ vslot, int n_virtual,
                                                       const int32_t* __restrict__ col,
                                                       const float* __restrict__ val, const float* __restrict__ X,
                                                       int64_t ldx, float* __restrict__ Y, int64_t ldy,
                                                       float* __restrict__ partial, int d, const SpmmAdd add) {
    constexpr int NPAR = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_virtual) return;
    const int sub = lane / LPR, cl = lane % LPR;
    const int64_t beg = vbeg[v], end = vend[v];
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int64_t j = beg + sub;
    for (; j + 3 * NPAR < end; j += 4 * NPAR) {
        const int c0 = col[j], c1 = col[j + NPAR], c2 = col[j + 2 * NPAR], c3 = col[j + 3 * NPAR];
        const float w0 = val[j], w1 = val[j + NPAR], w2 = val[j + 2 * NPAR], w3 = val[j + 3 * NPAR];
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(X + (int64_t)c0 * ldx + cl * 4);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(X + (int64_t)c1 * ldx + cl * 4);
        const f32x4 x2 = *reinterpret_cast<const f32x4*>(X + (int64_t)c2 * ldx + cl * 4);
        const f32x4 x3 = *reinterpret_cast<const f32x4*>(X + (int64_t)c3 * ldx + cl * 4);
        s0 += w0 * x0;
        s1 += w1 * x1;
        s2 += w2 * x2;
        s3 += w3 * x3;
    }
    for (; j < end; j += NPAR) {
        const int c0 = col[j];
        const float w0 = val[j];
        s0 += w0 * *reinterpret_cast<const f32x4*>(X + (int64_t)c0 * ldx + cl * 4);
    }
    f32x4 s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
        s.x += __shfl_xor(s.x, o);
        s.y += __shfl_xor(s.y, o);
        s.z += __shfl_xor(s.z, o);
        s.w += __shfl_xor(s.w, o);
    }
    if (sub == 0) {
        const int slot = vslot[v];
        if (slot >= 0) {
            *reinterpret_cast<f32x4*>(partial + (int64_t)slot * d + cl * 4) = s;
        } else {
            const int r = vrow[v];
            for (int k = 0; k < add.n; ++k) s += *reinterpret_cast<const f32x4*>(add.p[k] + (int64_t)r * add.ld + cl * 4);
            *reinterpret_cast<f32x4*>(Y + (int64_t)r * ldy + cl * 4) = s * add.scale;
        }
    }
}

// Short rows (the bulk of a user-item graph: median degree ~ 20): LPR lanes own one whole row, so a wave
// works on 64/LPR rows at once.  With one row per wave the kernel is bound by the dependent chain
// vptr -> col/val -> gather -> store of 50 k+ tiny waves (measured 39 us for the user half of the Yelp graph);
// here each lane group fetches up to LPR (col, val) pairs with one coalesced load, broadcasts them with
// group-local shuffles and keeps 4 gathers in flight.  Accumulation is in CSR order (deterministic).
template <int LPR>
__global__ __launch_bounds__(256) void spmm_short_kernel(const int64_t* __restrict__ vbeg, const int64_t* __restrict__ vend,
                                                         const int32_t* __restrict__ vrow, int n_short,
                                                         const int32_t* __restrict__ col,
                                                         const float* __restrict__ val, const float* __restrict__ X,
                                                         int64_t ldx, float* __restrict__ Y, int64_t ldy,
                                                         const SpmmAdd add) {
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPR, gl = lane % LPR;
    const int v = (blockIdx.x * 4 + (threadIdx.x >> 6)) * G + grp;
    if (v >= n_short) return;  // whole lane group leaves together
    const int64_t beg = vbeg[v], end = vend[v];
    const int r = vrow[v];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int64_t base = beg; base < end; base += LPR) {
        const bool have = base + gl < end;
        const int my_c = have ? col[base + gl] : 0;
        const float my_w = have ? val[base + gl] : 0.f;
        const int n = (int)min((int64_t)LPR, end - base);
        int j = 0;
        for (; j + 3 < n; j += 4) {
            const int c0 = __shfl(my_c, j, LPR), c1 = __shfl(my_c, j + 1, LPR), c2 = __shfl(my_c, j + 2, LPR),
                      c3 = __shfl(my_c, j + 3, LPR);
            const float w0 = __shfl(my_w, j, LPR), w1 = __shfl(my_w, j + 1, LPR), w2 = __shfl(my_w, j + 2, LPR),
                        w3 = __shfl(my_w, j + 3, LPR);
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(X + (int64_t)c0 * ldx + gl * 4);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(X + (int64_t)c1 * ldx + gl * 4);
            const f32x4 x2 = *reinterpret_cast<const f32x4*>(X + (int64_t)c2 * ldx + gl * 4);
            const f32x4 x3 = *reinterpret_cast<const f32x4*>(X + (int64_t)c3 * ldx + gl * 4);
            s += w0 * x0;
            s += w1 * x1;
            s += w2 * x2;
            s += w3 * x3;
        }
        for (; j < n; ++j) {
            const int c0 = __shfl(my_c, j, LPR);
            const float w0 = __shfl(my_w, j, LPR);
            s += w0 * *reinterpret_cast<const f32x4*>(X + (int64_t)c0 * ldx + gl * 4);
        }
    }
    for (int k = 0; k < add.n; ++k) s += *reinterpret_cast<const f32x4*>(add.p[k] + (int64_t)r * add.ld + gl * 4);
    *reinterpret_cast<f32x4*>(Y + (int64_t)r * ldy + gl * 4) = s * add.scale;
}

// generic fallback (any d): one neighbour at a time, lanes stride over columns
__global__ __launch_bounds__(256) void spmm_generic_kernel(const int64_t* __restrict__ vbeg, const int64_t* __restrict__ vend,
                                                           const int32_t* __restrict__ vrow,
                                                           const int32_t* __restrict__ vslot, int n_virtual,
                                                           const int32_t* __restrict__ col,
                                                           const float* __restrict__ val, const float* __restrict__ X,
                                                           int64_t ldx, float* __restrict__ Y, int64_t ldy,
                                                           float* __restrict__ partial, int d, const SpmmAdd add) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_virtual) return;
    const int64_t beg = vbeg[v], end = vend[v];
    const int slot = vslot[v], r = vrow[v];
    for (int c0 = lane; c0 < d; c0 += 64) {
        float s = 0.f;
        for (int64_t j = beg; j < end; ++j) s += val[j] * X[(int64_t)col[j] * ldx + c0];
        if (slot >= 0) {
            partial[(int64_t)slot * d + c0] = s;
        } else {
            for (int k = 0; k < add.n; ++k) s += add.p[k][(int64_t)r * add.ld + c0];
            Y[(int64_t)r * ldy + c0] = s * add.scale;
        }
    }
}

// rows that were split: Y[r] = (sum of their partial slots + addends) * scale.  One workgroup per split
// row: wave w adds slots w, w+4, ... (independent loads, 4 in flight), the four partial sums are added in
// wave order -> fixed summation order, no atomics.
__global__ __launch_bounds__(256) void spmm_combine_kernel(const int32_t* __restrict__ lrow,
                                                           const int32_t* __restrict__ lptr, int n_long,
                                                           const float* __restrict__ partial, int d,
                                                           float* __restrict__ Y, int64_t ldy, const SpmmAdd add) {
    extern __shared__ float comb[];  // [4][d]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = blockIdx.x;
    const int r = lrow[l];
    const int beg = lptr[l], end = lptr[l + 1];
    for (int c0 = lane; c0 < d; c0 += 64) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = beg + wave;
        for (; k + 12 < end; k += 16) {
            s0 += partial[(int64_t)k * d + c0];
            s1 += partial[(int64_t)(k + 4) * d + c0];
            s2 += partial[(int64_t)(k + 8) * d + c0];
            s3 += partial[(int64_t)(k + 12) * d + c0];
        }
        for (; k < end; k += 4) s0 += partial[(int64_t)k * d + c0];
        comb[wave * d + c0] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (wave == 0) {
        for (int c0 = lane; c0 < d; c0 += 64) {
            float s = ((comb[c0] + comb[d + c0]) + comb[2 * d + c0]) + comb[3 * d + c0];
            for (int k = 0; k < add.n; ++k) s += add.p[k][(int64_t)r * add.ld + c0];
            Y[(int64_t)r * ldy + c0] = s * add.scale;
        }
    }
}

// ---- ranking metrics (reference evaluate_utils.py:6-52) -----------------------------------------------------------
// One thread per user walks its K predicted items once, in rank order, exactly as the reference's inner loop does
// for every N (hits, dcg += 1/log2(j+2), ideal dcg over min(|GT|, N) ranks, reciprocal rank of the first hit), and
// emits the four per-user terms whenever j+1 reaches one of the requested cut-offs.  float64 throughout, additions
// in the reference's order: the per-user terms are bit-identical to the Python loop; the host adds them up in user
// order.  Ground-truth rows must have sorted column indices (binary search).
constexpr int TOPN_MAX = 8;
struct TopNList {
    int n;
    int v[TOPN_MAX];
};

__global__ __launch_bounds__(256) void topn_metrics_kernel(const int64_t* __restrict__ pred, int64_t ldp, int U,
                                                           const int64_t* __restrict__ gt_indptr,
                                                           const int32_t* __restrict__ gt_idx, const TopNList tn,
                                                           double* __restrict__ out) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= U) return;
    const int64_t beg = gt_indptr[u], end = gt_indptr[u + 1];
    const int64_t len = end - beg;
    double* o = out + (int64_t)u * tn.n * 4;
    if (len == 0) {
        for (int k = 0; k < tn.n * 4; ++k) o[k] = 0.0;
        return;
    }
    int hits = 0, next = 0;
    int64_t left = len;
    double dcg = 0.0, idcg = 0.0, mrr = 0.0;
    bool first = true;
    const int K = tn.v[tn.n - 1];
    for (int j = 0; j < K; ++j) {
        const int64_t item = pred[(int64_t)u * ldp + j];
        int64_t lo = beg, hi = end;  // lower bound of `item` in the sorted row
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)gt_idx[mid] < item) lo = mid + 1;
            else hi = mid;
        }
        const double gain = 1.0 / log2((double)(j + 2));
        if (lo < end && (int64_t)gt_idx[lo] == item) {
            dcg += gain;
            if (first) {
                mrr = 1.0 / ((double)j + 1.0);
                first = false;
            }
            ++hits;
        }
        if (left > 0) {
            idcg += gain;
            --left;
        }
        if (j + 1 == tn.v[next]) {
            const double N = (double)tn.v[next];
            o[next * 4 + 0] = (double)hits / N;
            o[next * 4 + 1] = (double)hits / (double)len;
            o[next * 4 + 2] = idcg != 0.0 ? dcg / idcg : 0.0;
            o[next * 4 + 3] = mrr;
            ++next;
        }
    }
}

}  // namespace

extern "C" {

int gdmcf_topk_masked_f32(const float* pred, int64_t ldp, int B, int I, const int64_t* mask_indptr,
                          const int32_t* mask_indices, int k, int64_t* idx_out, float* val_out, void* stream) {
    GD_CHECK_SHAPE(B > 0 && I > 0 && ldp >= I, "topk: bad shape");
    GD_CHECK_SHAPE(k >= 1 && k <= I, "topk: k out of range (selected index k out of range)");
    GD_CHECK_ARG(k <= 2048, "topk: k > 2048 unsupported");
    GD_CHECK_ARG((mask_indptr == nullptr) == (mask_indices == nullptr), "topk: mask indptr/indices mismatch");
    int KP = 2;
    while (KP < k) KP <<= 1;
    const size_t lds_base = (size_t)KP * 8 + (256 + 24) * 4 + (size_t)((I + 31) / 32) * 4;
    GdProfScope prof(9, 4.0 * B * (double)I, (hipStream_t)stream);  // (both launches)
    // the fast path (topk_fast_kernel): rows up to 40 960 wide, k <= 256 -- followed by topk_kernel for the rows it marks
    static const int fast_on = getenv("GDMCF_TOPK_FAST") ? atoi(getenv("GDMCF_TOPK_FAST")) : 1;  // 0: the radix-select kernel only
    const bool fast = fast_on && k <= 256 && I <= 1024 * 40 && k <= I;
    if (fast) {
        const size_t lds_f = (size_t)TOPK_FAST_CAP * 8 + (1024 + 4) * 4 + (size_t)((I + 31) / 32) * 4;
        if (I <= 1024 * 8)
            hipLaunchKernelGGL((topk_fast_kernel<1024, 8>), dim3(B), dim3(1024), lds_f, (hipStream_t)stream, pred, ldp, I, mask_indptr,
                               mask_indices, k, idx_out, val_out);
        else if (I <= 1024 * 34)
            hipLaunchKernelGGL((topk_fast_kernel<1024, 34>), dim3(B), dim3(1024), lds_f, (hipStream_t)stream, pred, ldp, I, mask_indptr,
                               mask_indices, k, idx_out, val_out);
        else
            hipLaunchKernelGGL((topk_fast_kernel<1024, 40>), dim3(B), dim3(1024), lds_f, (hipStream_t)stream, pred, ldp, I, mask_indptr,
                               mask_indices, k, idx_out, val_out);
    }
    const int redo = fast ? 1 : 0;
    // (behind the fast path the marked rows are rare: the variant that does not stage the row needs 6 KB of LDS instead of 140 and
    // its workgroups -- all but the marked ones return at once -- are dispatched many per CU)
    const bool stage = !redo && lds_base + (size_t)I * 4 <= 150 * 1024;
    const size_t lds = lds_base + (stage ? (size_t)I * 4 : 0);
    if (lds > 160 * 1024) {
        gdmcf_set_error("topk: row width %d needs %zu B of LDS (> 160 KiB)", I, lds);
        return GDMCF_E_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_kernel<true, 1024>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_kernel<false, 1024>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            gdmcf_set_error("topk: hipFuncSetAttribute failed");
            return GDMCF_E_HIP;
        }
        attr_set = true;
    }
    if (stage)
        hipLaunchKernelGGL((topk_kernel<true, 1024>), dim3(B), dim3(1024), lds, (hipStream_t)stream, pred, ldp, I, mask_indptr,
                           mask_indices, k, KP, idx_out, val_out, redo);
    else
        hipLaunchKernelGGL((topk_kernel<false, 1024>), dim3(B), dim3(1024), lds, (hipStream_t)stream, pred, ldp, I, mask_indptr,
                           mask_indices, k, KP, idx_out, val_out, redo);
    return gd_launch_status("topk");
}

int gdmcf_spmm_csr_f32(const int64_t* vbeg, const int64_t* vend, const int32_t* vrow, const int32_t* vslot, int n_virtual, int n_short,
                       const int32_t* lrow, const int32_t* lptr, int n_long, const int32_t* col, const float* val,
                       int n_rows, const float* X, int64_t ldx, int d, float* Y, int64_t ldy, float* partial_ws,
                       const float* const* addends_host, int n_add, int64_t ld_add, float scale, double alg_bytes,
                       void* stream) {
    GD_CHECK_SHAPE(n_rows > 0 && n_virtual > 0 && d > 0 && ldx >= d && ldy >= d, "spmm: bad shape");
    GD_CHECK_ARG(n_add >= 0 && n_add <= SPMM_MAX_ADD && (n_add == 0 || (addends_host && ld_add >= d)), "spmm: bad addends");
    GD_CHECK_ARG(n_long == 0 || (lrow && lptr && partial_ws), "spmm: split rows need lrow/lptr/partial_ws");
    hipStream_t s = (hipStream_t)stream;
    SpmmAdd add = {};
    add.n = n_add; add.ld = ld_add; add.scale = scale;
    bool add_al = true;
    for (int k = 0; k < n_add; ++k) {
        add.p[k] = addends_host[k];
        add_al = add_al && gd_aligned16(add.p[k]);
    }
    GD_CHECK_ARG(n_short >= 0 && n_short <= n_virtual, "spmm: n_short out of range");
    const dim3 block(256);
    const bool vec = (d % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && gd_aligned16(X) && gd_aligned16(Y) &&
                     (n_add == 0 || ((ld_add % 4 == 0) && add_al)) && (partial_ws == nullptr || gd_aligned16(partial_ws));
    const int lpr = d / 4;
    const bool pow2 = vec && (lpr == 2 || lpr == 4 || lpr == 8 || lpr == 16 || lpr == 32 || lpr == 64);
    if (!pow2 && n_short > 0) {
        gdmcf_set_error("spmm: the short-row path needs d in {8,16,32,64,128,256}; build the plan with n_short = 0");
        return GDMCF_E_ARG;
    }
    {
        GdProfScope prof(8, alg_bytes, s);
        if (n_short > 0) {
#define GD_SPMM_SHORT(L) hipLaunchKernelGGL(spmm_short_kernel<L>, dim3(gd_cdiv(n_short, 4 * (64 / L))), block, 0, s, vbeg, vend, vrow, n_short, col, val, X, ldx, Y, ldy, add)
            if (lpr == 16) GD_SPMM_SHORT(16);
            else if (lpr == 8) GD_SPMM_SHORT(8);
            else if (lpr == 32) GD_SPMM_SHORT(32);
            else if (lpr == 64) GD_SPMM_SHORT(64);
            else if (lpr == 4) GD_SPMM_SHORT(4);
            else GD_SPMM_SHORT(2);
#undef GD_SPMM_SHORT
        }
        const int n_rest = n_virtual - n_short;
        const dim3 grid(gd_cdiv(n_rest > 0 ? n_rest : 1, 4));
        const int64_t* vbeg_r = vbeg + n_short;
        const int64_t* vend_r = vend + n_short;
        const int32_t* vrow_r = vrow + n_short;
        const int32_t* vslot_r = vslot + n_short;
#define GD_SPMM_LAUNCH(K) hipLaunchKernelGGL(K, grid, block, 0, s, vbeg_r, vend_r, vrow_r, vslot_r, n_rest, col, val, X, ldx, Y, ldy, partial_ws, d, add)
        if (n_rest > 0) {
            if (vec && lpr == 16) GD_SPMM_LAUNCH(spmm_vec_kernel<16>);
            else if (vec && lpr == 8) GD_SPMM_LAUNCH(spmm_vec_kernel<8>);
            else if (vec && lpr == 32) GD_SPMM_LAUNCH(spmm_vec_kernel<32>);
            else if (vec && lpr == 64) GD_SPMM_LAUNCH(spmm_vec_kernel<64>);
            else if (vec && lpr == 4) GD_SPMM_LAUNCH(spmm_vec_kernel<4>);
            else if (vec && lpr == 2) GD_SPMM_LAUNCH(spmm_vec_kernel<2>);
            else GD_SPMM_LAUNCH(spmm_generic_kernel);
        }
#undef GD_SPMM_LAUNCH
        if (n_long > 0)
            hipLaunchKernelGGL(spmm_combine_kernel, dim3(n_long), block, (size_t)4 * d * sizeof(float), s, lrow, lptr,
                               n_long, partial_ws, d, Y, ldy, add);
    }
    return gd_launch_status("spmm_csr");
}

int gdmcf_topn_metrics_f64(const int64_t* pred_idx, int64_t ldp, int U, const int64_t* gt_indptr, const int32_t* gt_indices,
                           const int* topN_host, int n_topn, double* out, void* stream) {
    GD_CHECK_SHAPE(U > 0 && n_topn >= 1 && n_topn <= TOPN_MAX, "topn_metrics: bad shape (1..8 cut-offs)");
    GD_CHECK_ARG(pred_idx && gt_indptr && gt_indices && topN_host && out, "topn_metrics: null pointer");
    TopNList tn;
    tn.n = n_topn;
    for (int i = 0; i < n_topn; ++i) {
        tn.v[i] = topN_host[i];
        GD_CHECK_ARG(tn.v[i] >= 1 && (i == 0 || tn.v[i] > tn.v[i - 1]), "topn_metrics: cut-offs must be ascending and >= 1");
    }
    GD_CHECK_SHAPE(ldp >= tn.v[n_topn - 1], "topn_metrics: fewer predicted items per user than the largest cut-off");
    hipLaunchKernelGGL(topn_metrics_kernel, dim3(gd_cdiv(U, 256)), dim3(256), 0, (hipStream_t)stream, pred_idx, ldp, U,
                       gt_indptr, gt_indices, tn, out);
    return gd_launch_status("topn_metrics");
}

}  // extern "C"
