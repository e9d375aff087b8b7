// Ablation / tuning harness for csrc/gemm_f32.hip (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DGD_PROBE_...] tools/gemm_probe.hip -o tools/gemm_probe_<variant>
// Runs the five Yelp-shape products of the training step and prints ms / TFLOP/s per product.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <algorithm>
#include <vector>

#include "../gdmcf_amd/csrc/gemm_f32.hip"

bool g_gd_prof_on = false;
int gd_gemm_bf16_launch(int, int, int, int, GdGemm&, hipStream_t) { return GDMCF_E_UNSUPPORTED; }  // f32 probe only
int gd_gemm_small_launch(int, int, int, GdGemm&, hipStream_t) { return GDMCF_E_UNSUPPORTED; }
int gd_gemm_split_launch(int, int, int, int, GdGemm&, hipStream_t) { return GDMCF_E_UNSUPPORTED; }
#include "../gdmcf_amd/csrc/gemm_dr.hip"
void gd_prof_begin(int, double, hipStream_t) {}
void gd_prof_end(hipStream_t) {}
void gdmcf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));            \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

int main(int argc, char** argv) {
    const int B = 400, I = 34395, H = 1000, E = 10, ldk = 34432, ldi = 34432;
    const int ldh = getenv("GD_LDH") ? atoi(getenv("GD_LDH")) : 1024;  // leading dimension of the hidden activations
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    float *xin, *W1, *W2, *h, *diff, *slab, *dW1, *dW2, *tgt, *rowpart;
    CK(hipMalloc(&xin, (size_t)B * ldk * 4));
    CK(hipMalloc(&W1, (size_t)H * (I + E) * 4));
    CK(hipMalloc(&W2, (size_t)I * H * 4));
    CK(hipMalloc(&h, (size_t)B * ldh * 4));
    CK(hipMalloc(&diff, (size_t)B * ldi * 4));
    CK(hipMalloc(&tgt, (size_t)B * I * 4));
    CK(hipMalloc(&slab, (size_t)64 * B * 1024 * 4));  // up to 64 splits
    CK(hipMalloc(&dW1, (size_t)H * (I + E) * 4));
    CK(hipMalloc(&dW2, (size_t)I * H * 4));
    CK(hipMalloc(&rowpart, (size_t)B * 2200 * 4));
    CK(hipMemset(rowpart, 0, (size_t)B * 2200 * 4));
    std::vector<float> init((size_t)I * H);
    srand(1);
    for (auto& v : init) v = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
    CK(hipMemcpy(W2, init.data(), (size_t)I * H * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W1, init.data(), (size_t)H * (I + E) * 4 > init.size() * 4 ? init.size() * 4 : (size_t)H * (I + E) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(xin, init.data(), (size_t)B * ldk * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(diff, init.data(), (size_t)B * ldi * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(tgt, init.data(), (size_t)B * I * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(h, init.data(), (size_t)B * ldh * 4, hipMemcpyHostToDevice));
    hipStream_t s = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int splits = getenv("GD_SPLITS") ? atoi(getenv("GD_SPLITS")) : 12;

    struct Case { const char* name; int la, lb, epi, cls; GdGemm g; };
    std::vector<Case> cases;
    {   // GEMM1: xin[B,I+E] * W1[H,I+E]^T, split-K slabs
        GdGemm g = {}; g.A = xin; g.lda = ldk; g.B = W1; g.ldb = I + E; g.M = B; g.N = H; g.K = I + E; g.splits = splits;
        g.C = slab; g.ldc = 1024; g.slab_stride = (int64_t)B * 1024; g.m_fastest = 1;
        cases.push_back({"gemm1_fwd ", GD_LAY_KC, GD_LAY_KC, GD_EPI_SLAB, 0, g});
    }
    {   // GEMM2 + loss
        GdGemm g = {}; g.A = h; g.lda = ldh; g.B = W2; g.ldb = H; g.M = B; g.N = I; g.K = H; g.splits = 1; g.m_fastest = 1;
        g.aux = tgt; g.ldaux = I; g.C = diff; g.ldc = ldi; g.rowpart = rowpart; g.ld_rowpart = 2200;
        cases.push_back({"gemm2_loss", GD_LAY_KC, GD_LAY_KC, GD_EPI_LOSS, 0, g});
    }
    for (int nn : {26112, 13056}) {   // the same product cut to 2045 / 1020 tiles of 80 x 64: two / one per SIMD on the hybrid kernel
        if (!getenv("GD_ROUNDS")) break;
        GdGemm g = {}; g.A = h; g.lda = ldh; g.B = W2; g.ldb = H; g.M = B; g.N = nn; g.K = H; g.splits = 1; g.m_fastest = 1;
        g.aux = tgt; g.ldaux = I; g.C = diff; g.ldc = ldi; g.rowpart = rowpart; g.ld_rowpart = 2200;
        cases.push_back({nn == 26112 ? "loss 2040t" : "loss 1020t", GD_LAY_KC, GD_LAY_KC, GD_EPI_LOSS, 0, g});
    }
    {   // dh = diff * W2
        GdGemm g = {}; g.A = diff; g.lda = ldi; g.B = W2; g.ldb = H; g.M = B; g.N = H; g.K = I; g.splits = splits;
        g.C = slab; g.ldc = 1024; g.slab_stride = (int64_t)B * 1024; g.m_fastest = 1;
        cases.push_back({"bwd_input ", GD_LAY_KC, GD_LAY_MC, GD_EPI_SLAB, 0, g});
    }
    {   // dW2 = diff^T * h
        GdGemm g = {}; g.A = diff; g.lda = ldi; g.B = h; g.ldb = ldh; g.M = I; g.N = H; g.K = B; g.splits = 1; g.m_fastest = 0;
        g.C = dW2; g.ldc = H;
        cases.push_back({"dW2 cls0  ", GD_LAY_MC, GD_LAY_MC, GD_EPI_STORE, 0, g});
#ifdef GD_STAMP
        g.rowpart = rowpart;  // stamp buffer (cleared below)
#endif
        cases.push_back({"dW2 cls1  ", GD_LAY_MC, GD_LAY_MC, GD_EPI_STORE, 1, g});
    }
    {   // dW1 = dhp^T * xin
        GdGemm g = {}; g.A = h; g.lda = ldh; g.B = xin; g.ldb = ldk; g.M = H; g.N = I + E; g.K = B; g.splits = 1; g.m_fastest = 1;
        g.C = dW1; g.ldc = I + E;
        cases.push_back({"dW1 cls1  ", GD_LAY_MC, GD_LAY_MC, GD_EPI_STORE, 1, g});
        cases.push_back({"dW1 cls0  ", GD_LAY_MC, GD_LAY_MC, GD_EPI_STORE, 0, g});
    }
    if (getenv("GD_DUMP")) {
        extern int g_gd_dr_force;
        g_gd_dr_force = 9;
        std::vector<float> hh((size_t)B * ldh, 0.f), ww((size_t)I * H), dbgh(2 * 65536);
        for (int m = 0; m < B; ++m)
            for (int k = 0; k < H; ++k) hh[(size_t)m * ldh + k] = (float)(k + 1);
        for (int n = 0; n < I; ++n)
            for (int k = 0; k < H; ++k) ww[(size_t)n * H + k] = (float)(1000 * (n % 64) + k);
        CK(hipMemcpy(W2, ww.data(), ww.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(h, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
        float* dbg;
        CK(hipMalloc(&dbg, dbgh.size() * 4));
        CK(hipMemset(dbg, 0, dbgh.size() * 4));
        GdGemm g = cases[1].g;
        g.aux2 = dbg;
        if (gd_gemm_launch(GD_LAY_KC, GD_LAY_KC, GD_EPI_LOSS, 0, g, s)) return 1;
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(dbgh.data(), dbg, dbgh.size() * 4, hipMemcpyDeviceToHost));
        for (int t = 0; t < 2; ++t) {
            const float* D0 = dbgh.data() + t * 65536;
            printf("tile %d: wave %g xcc %g\n", t ? 5 : 0, D0[100], D0[101]);
            int bad = 0;
            for (int lane = 0; lane < 64; ++lane) {
                const float* D = D0 + lane * 256;
                const int q = lane >> 4, r = lane & 15, j = lane >> 3, sl = lane & 7;
                const int rows[7] = {0, 1, 2, 3, 16, 17, 18};
                for (int u = 0; u < 7; ++u) {
                    for (int e = 0; e < 4; ++e)
                        if (D[u * 4 + e] != (float)(rows[u] + 4 * q + 1) && bad++ < 12) printf("  A4 slot %d lane %d e %d: %g (want %d)\n", u, lane, e, D[u * 4 + e], rows[u] + 4 * q + 1);
                    if (D[28 + u] != (float)(rows[u] + 4 * q + 1) && bad++ < 12) printf("  A1 slot %d lane %d: %g\n", u, lane, D[28 + u]);
                }
                for (int u = 1; u < 8; ++u)
                    for (int e = 0; e < 4; ++e) {
                        const float want = 1000.f * ((t ? 64 : 0) % 64 + 8 * u + j) + 32 + 4 * sl + e;
                        if (D[36 + u * 4 + e] != want && bad++ < 12) printf("  G[%d] lane %d e %d: %g (want %g)\n", u, lane, e, D[36 + u * 4 + e], want);
                    }
                for (int b = 0; b < 4; ++b)
                    for (int e = 0; e < 4; ++e) {
                        const float w0 = 1000.f * (16 * b + r) + 4 * q + e, w1 = w0 + 32, w2 = w0 + 16;
                        if (D[68 + b * 4 + e] != w0 && bad++ < 24) printf("  FB0(entry)[%d] lane %d e %d: %g (want %g)\n", b, lane, e, D[68 + b * 4 + e], w0);
                        if (D[104 + b * 4 + e] != w1 && bad++ < 24) printf("  FB0(after step 7)[%d] lane %d e %d: %g (want %g)\n", b, lane, e, D[104 + b * 4 + e], w1);
                        if (D[120 + b * 4 + e] != w2 && bad++ < 24) printf("  FB1(after step 7)[%d] lane %d e %d: %g (want %g)\n", b, lane, e, D[120 + b * 4 + e], w2);
                    }
            }
            printf("  %d mismatches\n", bad);
            printf("  LDS buffer 1 after step 7, row: 8 slots' first element (swizzled slots):\n");
            for (int row = 0; row < 64; row += 1) {
                printf("   row %2d:", row);
                for (int sl = 0; sl < 8; ++sl) printf(" %7g", D0[row * 256 + 136 + 4 * sl]);
                printf("\n");
            }
        }
        return 0;
    }
    if (getenv("GD_DIAG")) {  // which (row, k) of W2 does the hybrid kernel multiply?  one-hot activations, index-valued weights
        extern int g_gd_dr_force;
        g_gd_dr_force = 9;
        std::vector<float> hh((size_t)B * ldh, 0.f), ww((size_t)I * H), out((size_t)B * ldi);
        CK(hipMemset(tgt, 0, (size_t)B * I * 4));
        const int kss[6] = {0, 5, 16, 37, 70, 999};
        for (int mode = 0; mode < 2; ++mode) {
            for (int n = 0; n < I; ++n)
                for (int k = 0; k < H; ++k) ww[(size_t)n * H + k] = mode ? (float)k : (float)(n % 4096);
            CK(hipMemcpy(W2, ww.data(), ww.size() * 4, hipMemcpyHostToDevice));
            for (int ki = 0; ki < 6; ++ki) {
                std::fill(hh.begin(), hh.end(), 0.f);
                for (int m = 0; m < B; ++m) hh[(size_t)m * ldh + kss[ki]] = 1.f;
                CK(hipMemcpy(h, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
                GdGemm g = cases[1].g;
                if (gd_gemm_launch(GD_LAY_KC, GD_LAY_KC, GD_EPI_LOSS, 0, g, s)) return 1;
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(out.data(), diff, out.size() * 4, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (int m = 0; m < B; ++m)
                    for (int n = 0; n < I; ++n) {
                        const float e = mode ? (float)kss[ki] : (float)(n % 4096);
                        if (out[(size_t)m * ldi + n] != e) ++bad;
                    }
                printf("diag mode %d (W2 = %s) one-hot k = %d: %zu wrong\n", mode, mode ? "k" : "n%4096", kss[ki], bad);
                if (getenv("GD_DIAG_V")) {
                    printf("   row 0, n = 64..127:");
                    for (int n = 64; n < 128; ++n) printf(" %g", out[n]);
                    printf("\n   row 77, n = 0..63:");
                    for (int n = 0; n < 64; ++n) printf(" %g", out[(size_t)77 * ldi + n]);
                    printf("\n");
                }
            }
        }
        return 0;
    }
    {   // the output-layer product on the hybrid kernel (dr_hl_kernel) against the LDS-tiled kernel: every element of diff, row sums
        extern int g_gd_dr_force;
        Case c = cases[1];
        std::vector<float> d0((size_t)B * ldi), d1((size_t)B * ldi), rp((size_t)B * 2200);
        std::vector<double> rs0(B, 0.0), rs1(B, 0.0);
        for (int pass = 0; pass < 2; ++pass) {
            g_gd_dr_force = pass ? 9 : 0;
            CK(hipMemset(diff, 0xff, (size_t)B * ldi * 4));
            GdGemm g = c.g;
            if (gd_gemm_launch(c.la, c.lb, c.epi, c.cls, g, s)) return 1;
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(pass ? d1.data() : d0.data(), diff, (size_t)B * ldi * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(rp.data(), rowpart, rp.size() * 4, hipMemcpyDeviceToHost));
            for (int m = 0; m < B; ++m)
                for (int t = 0; t < g.tiles_n; ++t) (pass ? rs1 : rs0)[m] += rp[(size_t)m * g.ld_rowpart + t];
            printf("loss product pass %d: tiles %d x %d, ld_rowpart %d\n", pass, g.tiles_m, g.tiles_n, g.ld_rowpart);
        }
        size_t nbad = 0, nbits = 0;
        double maxd = 0;
        for (int m = 0; m < B; ++m)
            for (int n = 0; n < I; ++n) {
                const float a = d0[(size_t)m * ldi + n], b = d1[(size_t)m * ldi + n];
                if (memcmp(&a, &b, 4)) ++nbits;
                const double e = fabs((double)a - b);
                if (!(e <= 1e-5 * (1 + fabs(a)))) { if (nbad < 5) printf("  diff[%d][%d] = %g vs %g\n", m, n, a, b); ++nbad; }
                if (e > maxd) maxd = e;
            }
        double maxr = 0;
        for (int m = 0; m < B; ++m) maxr = fmax(maxr, fabs(rs0[m] - rs1[m]) / (1e-30 + fabs(rs0[m])));
        printf("hybrid vs LDS-tiled loss product: %zu elements differ beyond 1e-5, %zu not bit-identical, max |d| %.3g, max rel row-sum diff %.3g\n",
               nbad, nbits, maxd, maxr);
        if ((nbad || maxr > 1e-5) && !getenv("GD_NOCHECK")) return 2;
        g_gd_dr_force = getenv("GD_DR_FORCE") ? atoi(getenv("GD_DR_FORCE")) : -1;
    }
    for (auto& c : cases) {
        for (int w = 0; w < 150; ++w) {
            GdGemm g = c.g;
            if (gd_gemm_launch(c.la, c.lb, c.epi, c.cls, g, s)) return 1;
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s));
        GdGemm g;
        for (int r = 0; r < reps; ++r) {
            g = c.g;
            gd_gemm_launch(c.la, c.lb, c.epi, c.cls, g, s);
        }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double fl = 2.0 * c.g.M * c.g.N * c.g.K;
        printf("%s  %.4f ms  %.1f TF  (grid %d)\n", c.name, ms, fl / ms / 1e9, g.tiles_m * g.tiles_n * g.splits);
#ifdef GD_STAMP
        if (c.g.rowpart && c.epi == GD_EPI_STORE && c.cls == 1) {
            std::vector<unsigned long long> st(8 * 32 * 4);
            CK(hipMemcpy(st.data(), rowpart, st.size() * 8, hipMemcpyDeviceToHost));
            for (int slot = 0; slot < 8; ++slot) {
                double rd = 0, mf = 0, bar = 0, tot = 0;
                int n = 0;
                for (int it = 1; it < 24; ++it) {
                    const unsigned long long* a = &st[(slot * 32 + it) * 4];
                    const unsigned long long* nx = &st[(slot * 32 + it + 1) * 4];
                    if (!a[0] || !nx[0]) continue;
                    rd += (double)(a[1] - a[0]); mf += (double)(a[2] - a[1]); bar += (double)(a[3] - a[2]);
                    tot += (double)(nx[0] - a[0]); ++n;
                }
                if (n) printf("  stamps wg %4d: per k-step  lds-read %.0f  mfma-issue %.0f  barrier %.0f  total %.0f  (s_memtime ticks, %d steps)\n",
                              slot * 269, rd / n, mf / n, bar / n, tot / n, n);
            }
        }
#endif
    }
    return 0;
}
