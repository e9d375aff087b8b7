// LightGCN propagation, second-generation CSR SpMM (reference lightGCN.py:184-189): ONE launch per layer over a
// host-built, statically balanced schedule (gdmcf_amd/lightgcn.py:spmm_bundle_plan).
//
// Why: with the gathered table confined to 0.5 MB (every gather an L2 hit) the first-generation kernels
// (topk_spmm.hip: short rows / long pieces / combine, three launches) still took 43 us per Yelp-shape layer
// (tools/spmm_ceiling.py) -- the time was in the kernels' shape, not in the cache misses: four rows of unrelated
// length shared a wave (the wave runs for the longest), every wave lived for four rows, three launches ramped up and
// drained one after the other.  Here
//   * rows of at most s_max nonzeros are sorted by length and bundled G = 64/LPR to a wave-step (LPR = d/4 lanes own
//     one 16-byte slice of a row of X each): equal lengths, no idle lane groups;
//   * longer rows are cut into pieces (<= piece nonzeros) that one wave gathers G neighbours at a time;
//   * every wave owns a CONTIGUOUS run of pieces + bundles of equal total cost, fixed by the host: 16-32 resident
//     waves per CU, all started at once, no second round of workgroups, no tail launch; descriptors and the
//     (col, val) batches of the next bundle are fetched while the current one gathers;
//   * block b runs on XCD b % 8 (round-robin dispatch): the schedule sorts work by mean column, so each XCD's 4 MiB L2
//     serves one slice of the gathered table (pieces of hub rows are column-local because CSR rows are sorted).
// Accumulation order is fixed by the schedule -> bit-identical run to run; rows cut into several pieces write
// float partials that spmm_combine adds in slot order (no atomics).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int SPMM_MAX_ADD = 8;
struct SpmmAdd {
    const float* p[SPMM_MAX_ADD];
    int n;
    int64_t ld;
    float scale;
};

struct SpmmBundlePlan {
    const int32_t* wdesc;  // [n_waves][4] = first/last+1 piece, first/last+1 bundle of the wave
    int waves_per_class;   // n_waves / 8; block b serves class b % 8
    const int64_t* lbeg;   // pieces: first nonzero, length, row, partial slot (-1 = the row is whole)
    const int32_t* llen;
    const int32_t* lrow;
    const int32_t* lslot;
    const int64_t* sbeg;  // bundles: G entries each -- first nonzero, length, row (-1 = padding)
    const int32_t* slen;
    const int32_t* srow;
    const int32_t* smax;  // [n_bundles] longest row of the bundle
};

// Row c of X, this lane's 16-byte slice.  WIDE = false: the whole table is addressable with a 32-bit byte offset from
// a wave-uniform base (one v_mad_u32 per gather instead of 64-bit address arithmetic).
template <bool WIDE, bool NT>
__device__ __forceinline__ f32x4 gather_row(const float* __restrict__ X, int64_t ldx, uint32_t ldx_bytes, int c, int gl) {
    const f32x4* p;
    if (WIDE) {
        p = reinterpret_cast<const f32x4*>(X + (int64_t)c * ldx + gl * 4);
    } else {
        const uint32_t off = (uint32_t)c * ldx_bytes + (uint32_t)gl * 16u;
        p = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(X) + off);
    }
    return NT ? __builtin_nontemporal_load(p) : *p;  // NT: rows nobody gathers again soon must not push the hot ones out of L2
}

// nk steps of one (col, val) batch held one entry per lane: step k gathers, for every lane group, the entry at lane
// k * KSTRIDE + goff (S bundles: KSTRIDE 1, goff = g*LPR; pieces: KSTRIDE G, goff = g), UN gathers back to back.
template <int LPR, int UN, bool WIDE, bool NT, int KSTRIDE>
__device__ __forceinline__ void gather_steps(f32x4& acc, int c_cur, float w_cur, int nk, int goff, int gl,
                                             const float* __restrict__ X, int64_t ldx, uint32_t ldxb) {
    for (int k = 0; k < nk; k += UN) {
        f32x4 x[UN];
        float wt[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int src = (k + u) * KSTRIDE + goff;  // < 64: k + u < LPR
            const int c = __shfl(c_cur, src);
            wt[u] = __shfl(w_cur, src);
            x[u] = gather_row<WIDE, NT>(X, ldx, ldxb, c, gl);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += wt[u] * x[u];
    }
}

// The loops below are branch-free inside a bundle / piece: entries past the end of a row are read from the row's last
// nonzero (clamped index: a neighbour the row really has) with weight 0, so every lane group issues the same UN
// back-to-back gathers per step and the compiler keeps counted waits (predicated loads became a branch and a full
// vmcnt(0) drain around every single gather: 13 TB/s of L2-resident gathers instead of the 32 TB/s the bare
// instruction sustains, tools/gather_probe.hip).
template <int LPR, int UNR, bool WIDE>
__global__ __launch_bounds__(256) void spmm_bundle_kernel(const SpmmBundlePlan pl, const int32_t* __restrict__ col,
                                                          const float* __restrict__ val, int64_t nnz,
                                                          const float* __restrict__ X, int64_t ldx, float* __restrict__ Y,
                                                          int64_t ldy, float* __restrict__ partial, int d, const SpmmAdd add) {
    constexpr int G = 64 / LPR;                 // rows of X gathered per wave instruction
    constexpr int UN = LPR >= UNR ? UNR : LPR;  // gathers in flight per wave (x 1 KiB)
    const int lane = threadIdx.x & 63;
    const int g = lane / LPR, gl = lane % LPR;
    const uint32_t ldxb = (uint32_t)ldx * 4u;
    const int w = (blockIdx.x & 7) * pl.waves_per_class + (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const int l0 = __builtin_amdgcn_readfirstlane(pl.wdesc[4 * w]);
    const int l1 = __builtin_amdgcn_readfirstlane(pl.wdesc[4 * w + 1]);
    const int s0 = __builtin_amdgcn_readfirstlane(pl.wdesc[4 * w + 2]);
    const int s1 = __builtin_amdgcn_readfirstlane(pl.wdesc[4 * w + 3]);
    const int64_t last = nnz - 1;

    // ---- pieces of long rows: the wave's 64 lanes fetch 64 (col, val) pairs, group g gathers entries k*G + g.
    // Software pipeline over pieces: descriptor two pieces ahead, first (col, val) batch one piece ahead (a piece that
    // waits for its own descriptor, then for its own batch, then for its gathers costs ~4 us of latency on its own). ----
    if (l0 < l1) {
        int64_t pbegA, pbegB, pbegC;
        int plenA, plenB, plenC, prowA, prowB, prowC, pslotA, pslotB, pslotC;
#define GD_PDESC(p, beg, len, row, slot) \
    do {                                 \
        beg = pl.lbeg[(p)];              \
        len = pl.llen[(p)];              \
        row = pl.lrow[(p)];              \
        slot = pl.lslot[(p)];            \
    } while (0)
        // lane `lane` holds entry base + lane of the piece, clamped to its last nonzero (weight 0 there)
#define GD_PBATCH(beg, len, base, c, wv)                                  \
    do {                                                                  \
        const int64_t j_ = beg + (int64_t)min((base) + lane, len - 1);    \
        c = col[j_];                                                      \
        wv = val[j_];                                                     \
        wv = (base) + lane < len ? wv : 0.f;                              \
    } while (0)
        GD_PDESC(l0, pbegA, plenA, prowA, pslotA);
        GD_PDESC(min(l0 + 1, l1 - 1), pbegB, plenB, prowB, pslotB);
        int pcA;
        float pwA;
        GD_PBATCH(pbegA, plenA, 0, pcA, pwA);
        for (int p = l0; p < l1; ++p) {
            GD_PDESC(min(p + 2, l1 - 1), pbegC, plenC, prowC, pslotC);
            int pcB;
            float pwB;
            GD_PBATCH(pbegB, plenB, 0, pcB, pwB);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int len = __builtin_amdgcn_readfirstlane(plenA);  // >= 1
            int c_cur = pcA;
            float w_cur = pwA;
            for (int base = 0; base < len; base += 64) {
                const int n = min(64, len - base);
                int c_nxt = 0;
                float w_nxt = 0.f;
                if (base + 64 < len) GD_PBATCH(pbegA, plenA, base + 64, c_nxt, w_nxt);  // uniform branch
                gather_steps<LPR, UN, WIDE, false, G>(acc, c_cur, w_cur, (n + G - 1) / G, g, gl, X, ldx, ldxb);
                c_cur = c_nxt;
                w_cur = w_nxt;
            }
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) {
                acc.x += __shfl_xor(acc.x, o);
                acc.y += __shfl_xor(acc.y, o);
                acc.z += __shfl_xor(acc.z, o);
                acc.w += __shfl_xor(acc.w, o);
            }
            const int row_st = prowA, slot_st = pslotA;
            pbegA = pbegB; plenA = plenB; prowA = prowB; pslotA = pslotB;
            pbegB = pbegC; plenB = plenC; prowB = prowC; pslotB = pslotC;
            pcA = pcB; pwA = pwB;
            asm volatile("" : "+v"(pbegA), "+v"(plenA), "+v"(prowA), "+v"(pslotA), "+v"(pbegB), "+v"(plenB), "+v"(prowB), "+v"(pslotB), "+v"(pcA), "+v"(pwA));
            if (g == 0) {
                if (slot_st >= 0) {
                    *reinterpret_cast<f32x4*>(partial + (int64_t)slot_st * d + gl * 4) = acc;
                } else {
                    for (int k = 0; k < add.n; ++k) acc += *reinterpret_cast<const f32x4*>(add.p[k] + (int64_t)row_st * add.ld + gl * 4);
                    *reinterpret_cast<f32x4*>(Y + (int64_t)row_st * ldy + gl * 4) = acc * add.scale;
                }
            }
        }
#undef GD_PDESC
#undef GD_PBATCH
    }

    // ---- bundles of G short rows of (nearly) equal length: group g owns row g of the bundle ----
    if (s0 >= s1) return;
    int64_t begA, begB, begC;
    int lenA, lenB, lenC, rowA, rowB, rowC, mxA, mxB, mxC;
#define GD_DESC(b, beg, len, row, mx)          \
    do {                                       \
        const int e_ = (b) * G + g;            \
        beg = pl.sbeg[e_];                     \
        len = pl.slen[e_];                     \
        row = pl.srow[e_];                     \
        mx = pl.smax[(b)];                     \
    } while (0)
    // (col, val) batch `base` of a row: lane gl holds entry base + gl, clamped to the row's last nonzero (weight 0 there);
    // an empty row reads some valid nonzero of the matrix with weight 0
#define GD_BATCH(beg, len, base, c, wv)                                                  \
    do {                                                                                 \
        const int64_t j_ = min(beg + (int64_t)min((base) + gl, max(len - 1, 0)), last);  \
        c = col[j_];                                                                     \
        wv = val[j_];                                                                    \
        wv = (base) + gl < len ? wv : 0.f;                                               \
    } while (0)
    GD_DESC(s0, begA, lenA, rowA, mxA);
    GD_DESC(min(s0 + 1, s1 - 1), begB, lenB, rowB, mxB);
    int cA;
    float wA;
    GD_BATCH(begA, lenA, 0, cA, wA);
    for (int b = s0; b < s1; ++b) {
        GD_DESC(min(b + 2, s1 - 1), begC, lenC, rowC, mxC);  // descriptor two bundles ahead
        int cB;                                              // first (col, val) batch one bundle ahead
        float wB;
        GD_BATCH(begB, lenB, 0, cB, wB);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int mxr = __builtin_amdgcn_readfirstlane(mxA);
        const int mx = mxr & 0x3FFFFFFF;
        const bool stream = (mxr >> 30) & 1;  // the bundle gathers mostly rows that are rarely gathered (host plan)
        int c_cur = cA;
        float w_cur = wA;
        for (int base = 0; base < mx; base += LPR) {
            const int nk = min(LPR, mx - base);
            int c_nxt = 0;
            float w_nxt = 0.f;
            if (base + LPR < mx) GD_BATCH(begA, lenA, base + LPR, c_nxt, w_nxt);  // uniform branch, rows > LPR only
            if (stream) gather_steps<LPR, UN, WIDE, true, 1>(acc, c_cur, w_cur, nk, g * LPR, gl, X, ldx, ldxb);
            else gather_steps<LPR, UN, WIDE, false, 1>(acc, c_cur, w_cur, nk, g * LPR, gl, X, ldx, ldxb);
            c_cur = c_nxt;
            w_cur = w_nxt;
        }
        // rotate the prefetched state first (everything loaded so far has arrived: no wait), store last -- a rotation
        // behind the store made the compiler drain the store (vmcnt(0)) at the end of every bundle
        const int row_st = rowA;
        if (lenA == 0) acc = f32x4{0.f, 0.f, 0.f, 0.f};  // nothing gathered: not even 0 * (a non-finite stranger)
        begA = begB; lenA = lenB; rowA = rowB; mxA = mxB;
        begB = begC; lenB = lenC; rowB = rowC; mxB = mxC;
        cA = cB; wA = wB;
        // pin the rotated values here: the compiler otherwise sinks the moves / selects below the store
        asm volatile("" : "+v"(begA), "+v"(lenA), "+v"(rowA), "+v"(mxA), "+v"(begB), "+v"(lenB), "+v"(rowB), "+v"(mxB), "+v"(cA), "+v"(wA));
        if (row_st >= 0) {
            for (int k = 0; k < add.n; ++k) acc += *reinterpret_cast<const f32x4*>(add.p[k] + (int64_t)row_st * add.ld + gl * 4);
            *reinterpret_cast<f32x4*>(Y + (int64_t)row_st * ldy + gl * 4) = acc * add.scale;
        }
    }
#undef GD_DESC
#undef GD_BATCH
}

// ---------------------------------------------------------------------------------------------------------------------
// Third generation: the wave streams its nonzeros.  The schedule is the one above, but the host also re-orders the
// (col, val) pairs into the order the waves gather them (gdmcf_amd/lightgcn.py:spmm_stream_pack): a wave reads ONE
// contiguous run, 64 entries (= LPR steps of G gathers) per load, whatever rows they belong to, and a unit (a piece or a
// bundle) is just "so many steps, then write these rows".  Why: vmcnt retires in order, so a load that goes to HBM (the
// col/val of the next row, a descriptor) holds back every younger L2-hit gather behind it -- a wave that chases row
// pointers pays an HBM latency per row (measured: 2.4 us per 40-nonzero piece).  Here the wave touches the lines of its
// run and of its descriptors once when it starts (one HBM round trip for everything), after which every load it waits
// for is an L2 hit.
// ---------------------------------------------------------------------------------------------------------------------
typedef int i32x2 __attribute__((ext_vector_type(2)));

struct SpmmStreamPlan {
    const int32_t* wdesc;  // [n_waves][4] = first batch of the wave's run, its batches, first / last+1 unit
    int waves_per_class;
    const i32x2* cw;       // (col, float bits of val), 64 entries per batch
    const int32_t* ud;     // [n_units][DW]
};

template <int LPR, bool WIDE>
__global__ __launch_bounds__(256) void spmm_stream_kernel(const SpmmStreamPlan pl, const float* __restrict__ X, int64_t ldx,
                                                          float* __restrict__ Y, int64_t ldy, float* __restrict__ partial,
                                                          int d, const SpmmAdd add, int dbg, long long* __restrict__ stamps) {
    constexpr int G = 64 / LPR;
    constexpr int UN = LPR >= 4 ? 4 : LPR;
    constexpr int DW = 1 + (G > 2 ? G : 2);
    const long long t_start = stamps ? wall_clock64() : 0;
    constexpr int GPB = LPR / UN;  // gather groups (UN steps each) per 64-entry batch
    const int lane = threadIdx.x & 63;
    const int g = lane / LPR, gl = lane % LPR;
    const uint32_t ldxb = (uint32_t)ldx * 4u;
    const int w = (blockIdx.x & 7) * pl.waves_per_class + (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const int sb = __builtin_amdgcn_readfirstlane(pl.wdesc[4 * w]);
    // (the packer gives every wave with units at least one batch -- spmm_stream_pack; the clamp keeps the pre-loads below
    // inside the run even for a hand-made descriptor with none)
    const int nb = max(__builtin_amdgcn_readfirstlane(pl.wdesc[4 * w + 1]), 1);
    const int u0 = __builtin_amdgcn_readfirstlane(pl.wdesc[4 * w + 2]);
    const int u1 = __builtin_amdgcn_readfirstlane(pl.wdesc[4 * w + 3]);
    if (u0 >= u1) return;
    const i32x2* __restrict__ run = pl.cw + (int64_t)sb * 64;
    const int32_t* __restrict__ udw = pl.ud + (int64_t)u0 * DW;
    // ---- warm-up: one dword of every 128-byte line (16 entries) of the first 32 batches and of the descriptors ----
    // (independent registers, consumed only at the very end: three loads in flight together, one HBM round trip)
    int t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    {
        const int lines = min(nb * 4, 128);
        const int* rl = reinterpret_cast<const int*>(run);
        t0 = rl[min(lane, lines - 1) * 32];
        t1 = rl[min(lane + 64, lines - 1) * 32];
        const int dlast = (u1 - u0) * DW - 1;
        t2 = udw[min(lane * 32, dlast)];
    }
    // the run and the results are streamed (non-temporal): an XCD's L2 is for the rows of X it gathers again and again
    i32x2 cur = __builtin_nontemporal_load(run + lane);
    i32x2 nxt = __builtin_nontemporal_load(run + min(1, nb - 1) * 64 + lane);
    int b = 0, q = 0;
    int hA = udw[0], aA = udw[1 + g], sA = udw[2];
    long long t_first = 0, t_pieces = 0;
    for (int u = u0; u < u1; ++u) {
        const int un = min(u + 1, u1 - 1) - u0;  // descriptor one unit ahead (L2 warm)
        int hB = hA, aB = aA, sB = sA;
        if (!(dbg & 4)) { hB = udw[un * DW]; aB = udw[un * DW + 1 + g]; sB = udw[un * DW + 2]; }
        const int hdr = __builtin_amdgcn_readfirstlane(hA);
        const int ngroups = hdr & 0x7FFFFFFF;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (stamps) {
            if (u == u0) t_first = wall_clock64();  // descriptors and the first batch have arrived
            if (hdr >= 0 && t_pieces == 0) t_pieces = wall_clock64();
        }
        for (int i = 0; i < ngroups; ++i) {
            f32x4 x[UN];
            float wt[UN];
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int src = (q * UN + j) * G + g;
                const int c = __shfl(cur.x, src);
                wt[j] = __int_as_float(__shfl(cur.y, src));
                x[j] = gather_row<WIDE, false>(X, ldx, ldxb, c, gl);
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) acc += wt[j] * x[j];
            if (++q == GPB) {  // batch used up: the next one is in registers, fetch the one after
                q = 0;
                ++b;
                cur = nxt;
                nxt = __builtin_nontemporal_load(run + (int64_t)min(b + 1, nb - 1) * 64 + lane);
                if ((b & 15) == 0)  // every 16 batches: touch the lines of batches b+16 .. b+31 (never waited for here)
                    t3 = reinterpret_cast<const int*>(run)[min((b + 16) * 4 + lane, nb * 4 - 1) * 32];
            }
        }
        if (hdr < 0 && !(dbg & 2)) {  // piece: add the lane groups up; group 0 writes the row or its partial slot
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) {
                acc.x += __shfl_xor(acc.x, o);
                acc.y += __shfl_xor(acc.y, o);
                acc.z += __shfl_xor(acc.z, o);
                acc.w += __shfl_xor(acc.w, o);
            }
        }
        const int a_st = aA, s_st = sA;
        hA = hB; aA = aB; sA = sB;
        asm volatile("" : "+v"(hA), "+v"(aA), "+v"(sA));  // keep the rotation above the stores
        if (dbg & 1) {
            if (acc.x == 1.2345f) Y[0] = acc.y;
        } else if (hdr < 0) {
            if (g == 0) {
                if (s_st >= 0) {
                    __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(partial + (int64_t)s_st * d + gl * 4));
                } else {
                    for (int k = 0; k < add.n; ++k) acc += *reinterpret_cast<const f32x4*>(add.p[k] + (int64_t)a_st * add.ld + gl * 4);
                    __builtin_nontemporal_store(acc * add.scale, reinterpret_cast<f32x4*>(Y + (int64_t)a_st * ldy + gl * 4));
                }
            }
        } else if (a_st >= 0) {  // bundle: lane group g owns row a_st (bit 30: the row is empty, discard what was gathered)
            const int r = a_st & 0x3FFFFFFF;
            if (a_st & 0x40000000) acc = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < add.n; ++k) acc += *reinterpret_cast<const f32x4*>(add.p[k] + (int64_t)r * add.ld + gl * 4);
            __builtin_nontemporal_store(acc * add.scale, reinterpret_cast<f32x4*>(Y + (int64_t)r * ldy + gl * 4));
        }
    }
    asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3));  // the touches are only waited for here
    if (stamps && lane == 0) {
        long long* o = stamps + (long long)w * 8;
        int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = t_start; o[1] = t_first; o[2] = t_pieces ? t_pieces : wall_clock64(); o[3] = wall_clock64();
        o[4] = u1 - u0; o[5] = nb; o[6] = xcc & 0xf; o[7] = blockIdx.x;
    }
}

// rows cut into several pieces: Y[r] = (sum of the partial slots in slot order + addends) * scale; one wave per row,
// LPR lanes per slot -> G slots per instruction, four instructions in flight; fixed order, no atomics
template <int LPR>
__global__ __launch_bounds__(256) void spmm_bundle_combine_kernel(const int32_t* __restrict__ crow, const int32_t* __restrict__ cptr,
                                                                  int n_cut, const float* __restrict__ partial, int d,
                                                                  float* __restrict__ Y, int64_t ldy, const SpmmAdd add) {
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, gl = lane % LPR;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= n_cut) return;
    const int r = crow[l], beg = cptr[l], end = cptr[l + 1];
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int k = beg + g;
    for (; k + 3 * G < end; k += 4 * G) {
        s0 += *reinterpret_cast<const f32x4*>(partial + (int64_t)k * d + gl * 4);
        s1 += *reinterpret_cast<const f32x4*>(partial + (int64_t)(k + G) * d + gl * 4);
        s2 += *reinterpret_cast<const f32x4*>(partial + (int64_t)(k + 2 * G) * d + gl * 4);
        s3 += *reinterpret_cast<const f32x4*>(partial + (int64_t)(k + 3 * G) * d + gl * 4);
    }
    for (; k < end; k += G) s0 += *reinterpret_cast<const f32x4*>(partial + (int64_t)k * d + gl * 4);
    f32x4 s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
        s.x += __shfl_xor(s.x, o);
        s.y += __shfl_xor(s.y, o);
        s.z += __shfl_xor(s.z, o);
        s.w += __shfl_xor(s.w, o);
    }
    if (g == 0) {
        for (int q = 0; q < add.n; ++q) s += *reinterpret_cast<const f32x4*>(add.p[q] + (int64_t)r * add.ld + gl * 4);
        *reinterpret_cast<f32x4*>(Y + (int64_t)r * ldy + gl * 4) = s * add.scale;
    }
}

template <int LPR>
void launch_bundle(int n_blocks, hipStream_t s, const SpmmBundlePlan& pl, const int32_t* col, const float* val, int64_t nnz,
                   const float* X, int64_t ldx, int n_x_rows, float* Y, int64_t ldy, float* partial, int d, const SpmmAdd& add,
                   const int32_t* crow, const int32_t* cptr, int n_cut) {
    static const int unr = getenv("GDMCF_SPMM_UN") ? atoi(getenv("GDMCF_SPMM_UN")) : 4;  // tuning knob
    const bool wide = (double)n_x_rows * (double)ldx * 4.0 >= 4294967296.0 || ldx * 4 >= (int64_t)1 << 31;
#define GD_K(U, W) hipLaunchKernelGGL((spmm_bundle_kernel<LPR, U, W>), dim3(n_blocks), dim3(256), 0, s, pl, col, val, nnz, X, ldx, Y, ldy, partial, d, add)
    if (wide) {
        if (unr == 8) GD_K(8, true);
        else GD_K(4, true);
    } else {
        if (unr == 8) GD_K(8, false);
        else GD_K(4, false);
    }
#undef GD_K
    if (n_cut > 0)
        hipLaunchKernelGGL(spmm_bundle_combine_kernel<LPR>, dim3(gd_cdiv(n_cut, 4)), dim3(256), 0, s, crow, cptr, n_cut, partial, d,
                           Y, ldy, add);
}

}  // namespace

extern "C" int gdmcf_spmm_bundled_f32(const int32_t* wdesc, int n_waves, const int64_t* lbeg, const int32_t* llen,
                                      const int32_t* lrow, const int32_t* lslot, int n_pieces, const int64_t* sbeg,
                                      const int32_t* slen, const int32_t* srow, const int32_t* smax, int n_bundles,
                                      const int32_t* crow, const int32_t* cptr, int n_cut, const int32_t* col,
                                      const float* val, int64_t nnz, int n_rows, int n_x_rows, const float* X, int64_t ldx, int d, float* Y, int64_t ldy,
                                      float* partial_ws, const float* const* addends_host, int n_add, int64_t ld_add,
                                      float scale, double alg_bytes, void* stream) {
    GD_CHECK_SHAPE(n_rows > 0 && n_x_rows > 0 && nnz > 0 && d > 0 && ldx >= d && ldy >= d, "spmm_bundled: bad shape");
    GD_CHECK_ARG(n_waves > 0 && n_waves % 32 == 0 && wdesc, "spmm_bundled: n_waves must be a positive multiple of 32");
    GD_CHECK_ARG(n_pieces >= 0 && n_bundles >= 0 && n_cut >= 0, "spmm_bundled: negative count");
    GD_CHECK_ARG(n_pieces == 0 || (lbeg && llen && lrow && lslot), "spmm_bundled: pieces without their arrays");
    GD_CHECK_ARG(n_bundles == 0 || (sbeg && slen && srow && smax), "spmm_bundled: bundles without their arrays");
    GD_CHECK_ARG(n_cut == 0 || (crow && cptr && partial_ws), "spmm_bundled: cut rows need crow/cptr/partial_ws");
    GD_CHECK_ARG(n_add >= 0 && n_add <= SPMM_MAX_ADD && (n_add == 0 || (addends_host && ld_add >= d)), "spmm_bundled: bad addends");
    const int lpr = d / 4;
    bool ok = (d % 4 == 0) && (lpr == 2 || lpr == 4 || lpr == 8 || lpr == 16 || lpr == 32 || lpr == 64) && (ldx % 4 == 0) &&
              (ldy % 4 == 0) && gd_aligned16(X) && gd_aligned16(Y) && (partial_ws == nullptr || gd_aligned16(partial_ws));
    SpmmAdd add = {};
    add.n = n_add;
    add.ld = ld_add;
    add.scale = scale;
    for (int k = 0; k < n_add; ++k) {
        add.p[k] = addends_host[k];
        ok = ok && gd_aligned16(add.p[k]) && (ld_add % 4 == 0);
    }
    if (!ok) {
        gdmcf_set_error("spmm_bundled: needs d in {8,16,32,64,128,256} and 16-byte aligned rows (use gdmcf_spmm_csr_f32 otherwise)");
        return GDMCF_E_UNSUPPORTED;
    }
    SpmmBundlePlan pl = {wdesc, n_waves / 8, lbeg, llen, lrow, lslot, sbeg, slen, srow, smax};
    hipStream_t s = (hipStream_t)stream;
    const int n_blocks = n_waves / 4;
    {
        GdProfScope prof(8, alg_bytes, s);
#define GD_GO(L) launch_bundle<L>(n_blocks, s, pl, col, val, nnz, X, ldx, n_x_rows, Y, ldy, partial_ws, d, add, crow, cptr, n_cut)
        if (lpr == 16) GD_GO(16);
        else if (lpr == 8) GD_GO(8);
        else if (lpr == 32) GD_GO(32);
        else if (lpr == 64) GD_GO(64);
        else if (lpr == 4) GD_GO(4);
        else GD_GO(2);
#undef GD_GO
    }
    return gd_launch_status("spmm_bundled");
}

static long long* g_spmm_stamps = nullptr;  // development aid: per-wave timestamps of the next launches (tools/spmm_waves.py)
static int g_spmm_stamp_waves = 0;

extern "C" int gdmcf_debug_spmm_stamps(int n_waves, long long* host_out) {
    // n_waves > 0, host_out == NULL: start recording (8 int64 per wave);  host_out != NULL: copy out and stop
    if (host_out) {
        if (!g_spmm_stamps) return GDMCF_E_ARG;
        hipError_t e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(host_out, g_spmm_stamps, (size_t)g_spmm_stamp_waves * 64, hipMemcpyDeviceToHost);
        (void)hipFree(g_spmm_stamps);
        g_spmm_stamps = nullptr;
        return e == hipSuccess ? GDMCF_OK : GDMCF_E_HIP;
    }
    if (g_spmm_stamps) (void)hipFree(g_spmm_stamps);
    g_spmm_stamps = nullptr;
    if (hipMalloc(&g_spmm_stamps, (size_t)n_waves * 64) != hipSuccess) return GDMCF_E_HIP;
    if (hipMemset(g_spmm_stamps, 0, (size_t)n_waves * 64) != hipSuccess) return GDMCF_E_HIP;
    g_spmm_stamp_waves = n_waves;
    return GDMCF_OK;
}

extern "C" int gdmcf_spmm_stream_f32(const int32_t* wdesc, int n_waves, const int32_t* cw, int64_t n_entries, const int32_t* ud,
                                     int n_units, const int32_t* crow, const int32_t* cptr, int n_cut, int n_rows, int n_x_rows,
                                     const float* X, int64_t ldx, int d, float* Y, int64_t ldy, float* partial_ws,
                                     const float* const* addends_host, int n_add, int64_t ld_add, float scale, double alg_bytes,
                                     void* stream) {
    GD_CHECK_SHAPE(n_rows > 0 && n_x_rows > 0 && d > 0 && ldx >= d && ldy >= d, "spmm_stream: bad shape");
    GD_CHECK_ARG(n_waves > 0 && n_waves % 32 == 0 && wdesc, "spmm_stream: n_waves must be a positive multiple of 32");
    GD_CHECK_ARG(n_units >= 0 && n_entries >= 0 && n_entries % 64 == 0 && n_cut >= 0, "spmm_stream: bad counts");
    GD_CHECK_ARG(n_units == 0 || (cw && ud), "spmm_stream: units without their arrays");
    GD_CHECK_ARG(n_cut == 0 || (crow && cptr && partial_ws), "spmm_stream: cut rows need crow/cptr/partial_ws");
    GD_CHECK_ARG(n_add >= 0 && n_add <= SPMM_MAX_ADD && (n_add == 0 || (addends_host && ld_add >= d)), "spmm_stream: bad addends");
    const int lpr = d / 4;
    bool ok = (d % 4 == 0) && (lpr == 2 || lpr == 4 || lpr == 8 || lpr == 16 || lpr == 32 || lpr == 64) && (ldx % 4 == 0) &&
              (ldy % 4 == 0) && gd_aligned16(X) && gd_aligned16(Y) && (partial_ws == nullptr || gd_aligned16(partial_ws)) &&
              ((reinterpret_cast<uintptr_t>(cw) & 7u) == 0);
    SpmmAdd add = {};
    add.n = n_add;
    add.ld = ld_add;
    add.scale = scale;
    for (int k = 0; k < n_add; ++k) {
        add.p[k] = addends_host[k];
        ok = ok && gd_aligned16(add.p[k]) && (ld_add % 4 == 0);
    }
    if (!ok) {
        gdmcf_set_error("spmm_stream: needs d in {8,16,32,64,128,256} and 16-byte aligned rows (use gdmcf_spmm_csr_f32 otherwise)");
        return GDMCF_E_UNSUPPORTED;
    }
    SpmmStreamPlan pl = {wdesc, n_waves / 8, reinterpret_cast<const i32x2*>(cw), ud};
    hipStream_t s = (hipStream_t)stream;
    const int n_blocks = n_waves / 4;
    const bool wide = (double)n_x_rows * (double)ldx * 4.0 >= 4294967296.0 || ldx * 4 >= (int64_t)1 << 31;
    static const int dbg = getenv("GDMCF_SPMM_DBG") ? atoi(getenv("GDMCF_SPMM_DBG")) : 0;  // ablation switches (tools/spmm_probe2.py)
    {
        GdProfScope prof(8, alg_bytes, s);
#define GD_GO(L)                                                                                                                        \
    do {                                                                                                                                \
        if (wide) hipLaunchKernelGGL((spmm_stream_kernel<L, true>), dim3(n_blocks), dim3(256), 0, s, pl, X, ldx, Y, ldy, partial_ws, d, add, dbg, g_spmm_stamps); \
        else hipLaunchKernelGGL((spmm_stream_kernel<L, false>), dim3(n_blocks), dim3(256), 0, s, pl, X, ldx, Y, ldy, partial_ws, d, add, dbg, g_spmm_stamps);     \
        if (n_cut > 0 && !(dbg & 8))                                                                                                                  \
            hipLaunchKernelGGL(spmm_bundle_combine_kernel<L>, dim3(gd_cdiv(n_cut, 4)), dim3(256), 0, s, crow, cptr, n_cut, partial_ws,  \
                               d, Y, ldy, add);                                                                                         \
    } while (0)
        if (lpr == 16) GD_GO(16);
        else if (lpr == 8) GD_GO(8);
        else if (lpr == 32) GD_GO(32);
        else if (lpr == 64) GD_GO(64);
        else if (lpr == 4) GD_GO(4);
        else GD_GO(2);
#undef GD_GO
    }
    return gd_launch_status("spmm_stream");
}
