// Ablation / tuning harness for csrc/gemm_f32.hip (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DGD_PROBE_...] tools/gemm_probe.hip -o tools/gemm_probe_<variant>
// Runs the five Yelp-shape products of the training step and prints ms / TFLOP/s per product.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
