// Library plumbing (errors, device info) and the host-side float64 schedule builder.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <vector>

#include <mutex>
#include <unordered_map>

#include "common.h"

static thread_local char g_err[512] = "";

void gdmcf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- event timing ----------------------------------------------------------------------------
bool g_gd_prof_on = false;
namespace {
struct ProfRec {
    hipEvent_t a, b;
    int tag;
    double work;
};
std::vector<ProfRec> g_prof_pool;   // events are created once and reused
size_t g_prof_used = 0;
}  // namespace

void gd_prof_begin(int tag, double work, hipStream_t s) {
    if (g_prof_used == g_prof_pool.size()) {
        if (g_prof_pool.size() >= 16384) return;
        ProfRec r;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        g_prof_pool.push_back(r);
    }
    ProfRec& r = g_prof_pool[g_prof_used];
    r.tag = tag;
    r.work = work;
    (void)hipEventRecord(r.a, s);
}

void gd_prof_end(hipStream_t s) {
    if (g_prof_used >= g_prof_pool.size()) return;
    (void)hipEventRecord(g_prof_pool[g_prof_used].b, s);
    ++g_prof_used;
}

// ---- bf16 shadows ---------------------------------------------------------------------------------------------
namespace {
std::mutex g_shadow_mu;
std::unordered_map<const void*, GdShadow> g_shadows;
}  // namespace

bool gd_shadow_lookup(const void* f32, GdShadow* out) {
    if (f32 == nullptr) return false;
    std::lock_guard<std::mutex> lk(g_shadow_mu);
    if (g_shadows.empty()) return false;
    auto it = g_shadows.find(f32);
    if (it == g_shadows.end()) return false;
    if (out) *out = it->second;
    return true;
}

thread_local int t_gd_last_gemm = 0;

extern "C" {

int gdmcf_bf16_shadow_set(const float* f32, void* bf16, int64_t rows, int64_t cols, int64_t ld_bf16) {
    GD_CHECK_ARG(f32 != nullptr && bf16 != nullptr, "bf16_shadow_set: null pointer");
    GD_CHECK_SHAPE(rows > 0 && cols > 0, "bf16_shadow_set: empty matrix");
    GD_CHECK_SHAPE(ld_bf16 % 64 == 0 && ld_bf16 >= cols, "bf16_shadow_set: ld_bf16 must be a multiple of 64 and >= cols");
    GD_CHECK_ARG((reinterpret_cast<uintptr_t>(bf16) & 15u) == 0, "bf16_shadow_set: shadow must be 16-byte aligned");
    std::lock_guard<std::mutex> lk(g_shadow_mu);
    g_shadows[f32] = GdShadow{bf16, ld_bf16, rows, cols};
    return GDMCF_OK;
}

int gdmcf_bf16_shadow_clear(const float* f32) {
    std::lock_guard<std::mutex> lk(g_shadow_mu);
    if (f32 == nullptr)
        g_shadows.clear();
    else
        g_shadows.erase(f32);
    return GDMCF_OK;
}

void* gdmcf_bf16_shadow_get(const float* f32) {
    GdShadow sh;
    return gd_shadow_lookup(f32, &sh) ? sh.p16 : nullptr;
}

int gdmcf_bf16_shadow_info(const float* f32, void** bf16, int64_t* rows, int64_t* cols, int64_t* ld_bf16) {
    GdShadow sh;
    if (!gd_shadow_lookup(f32, &sh)) return 0;
    if (bf16) *bf16 = sh.p16;
    if (rows) *rows = sh.rows;
    if (cols) *cols = sh.cols;
    if (ld_bf16) *ld_bf16 = sh.ld16;
    return 1;
}

int gdmcf_bf16_shadow_sync(const float* f32, int64_t ld, void* stream) {
    GdShadow sh;
    if (!gd_shadow_lookup(f32, &sh)) {
        gdmcf_set_error("bf16_shadow_sync: no shadow registered for %p", (const void*)f32);
        return GDMCF_E_ARG;
    }
    GD_CHECK_SHAPE(ld >= sh.cols, "bf16_shadow_sync: ld < cols");
    return gd_cast_bf16(f32, ld, sh.p16, sh.ld16, sh.rows, sh.cols, (hipStream_t)stream);
}

int gdmcf_prof_enable(int on) {
    g_gd_prof_on = (on == 1);
    if (on == 0) g_prof_used = 0;  // 2 = pause: stop recording, keep what was recorded
    return GDMCF_OK;
}

int gdmcf_prof_collect(int cap, int* tags, float* ms, double* work) {
    int n = 0;
    for (size_t i = 0; i < g_prof_used && n < cap; ++i) {
        ProfRec& r = g_prof_pool[i];
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
        tags[n] = r.tag;
        ms[n] = t;
        work[n] = r.work;
        ++n;
    }
    g_prof_used = 0;
    return n;
}

int gdmcf_version(void) { return 1; }

int gdmcf_debug_last_gemm(void) { return t_gd_last_gemm; }

const char* gdmcf_last_error(void) { return g_err; }

int gdmcf_device_info(int* n_cu, int* wave_size, char* arch_host, int arch_len) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        gdmcf_set_error("no HIP device");
        return GDMCF_E_HIP;
    }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (arch_host && arch_len > 0) {
        strncpy(arch_host, p.gcnArchName, (size_t)arch_len - 1);
        arch_host[arch_len - 1] = 0;
    }
    return GDMCF_OK;
}

// Restates, in float64 with the same operation order, reference
// models/gaussian_diffusion.py:109-159 (+ :1138-1163).  numpy.linspace is reproduced as
// y[i] = i*step + start with step = (stop-start)/(n-1) and y[n-1] = stop.
int gdmcf_schedule_build(int kind, double noise_scale, double noise_min, double noise_max, int T, int beta_fixed,
                         double* out) {
    GD_CHECK_SHAPE(T >= 1 && out != nullptr, "schedule_build: T < 1");
    std::vector<double> betas((size_t)T);
    if (kind == 0 || kind == 1) {
        const double start = noise_scale * noise_min, stop = noise_scale * noise_max;
        std::vector<double> lin((size_t)T);
        if (T == 1) {
            lin[0] = start;
        } else {
            const double step = (stop - start) / (double)(T - 1);
            for (int i = 0; i < T; ++i) lin[i] = (step == 0.0) ? ((double)i / (double)(T - 1)) * (stop - start) + start
                                                               : (double)i * step + start;
            lin[T - 1] = stop;
        }
        if (kind == 0) {
            betas = lin;
        } else {  // linear-var: alpha_bar = 1 - variance
            betas[0] = 1.0 - (1.0 - lin[0]);
            for (int i = 1; i < T; ++i) betas[i] = fmin(1.0 - (1.0 - lin[i]) / (1.0 - lin[i - 1]), 0.999);
        }
    } else if (kind == 2) {
        auto abar = [](double t) {
            const double c = cos((t + 0.008) / 1.008 * M_PI / 2);
            return c * c;
        };
        for (int i = 0; i < T; ++i) {
            const double t1 = (double)i / (double)T, t2 = (double)(i + 1) / (double)T;
            betas[i] = fmin(1.0 - abar(t2) / abar(t1), 0.999);
        }
    } else if (kind == 3) {
        for (int i = 0; i < T; ++i) betas[i] = 1.0 / (double)(T - i + 1);
    } else {
        gdmcf_set_error("unknown beta schedule kind %d", kind);
        return GDMCF_E_UNSUPPORTED;
    }
    if (beta_fixed) betas[0] = 0.00001;
    for (int i = 0; i < T; ++i) {
        if (!(betas[i] > 0.0 && betas[i] <= 1.0)) {
            gdmcf_set_error("betas out of range");
            return GDMCF_E_SHAPE;
        }
    }
    double* b = out;
    double* ac = out + (size_t)T;
    double* acp = out + (size_t)2 * T;
    double* acn = out + (size_t)3 * T;
    double* sac = out + (size_t)4 * T;
    double* s1m = out + (size_t)5 * T;
    double* l1m = out + (size_t)6 * T;
    double* srec = out + (size_t)7 * T;
    double* srecm1 = out + (size_t)8 * T;
    double* pv = out + (size_t)9 * T;
    double* plv = out + (size_t)10 * T;
    double* c1 = out + (size_t)11 * T;
    double* c2 = out + (size_t)12 * T;
    double run = 1.0;
    for (int i = 0; i < T; ++i) {
        b[i] = betas[i];
        run = (i == 0) ? (1.0 - betas[0]) : run * (1.0 - betas[i]);
        ac[i] = run;
    }
    for (int i = 0; i < T; ++i) {
        acp[i] = (i == 0) ? 1.0 : ac[i - 1];
        acn[i] = (i == T - 1) ? 0.0 : ac[i + 1];
        sac[i] = sqrt(ac[i]);
        s1m[i] = sqrt(1.0 - ac[i]);
        l1m[i] = log(1.0 - ac[i]);
        srec[i] = sqrt(1.0 / ac[i]);
        srecm1[i] = sqrt(1.0 / ac[i] - 1);
        pv[i] = b[i] * (1.0 - acp[i]) / (1.0 - ac[i]);
        c1[i] = b[i] * sqrt(acp[i]) / (1.0 - ac[i]);
        c2[i] = (1.0 - acp[i]) * sqrt(1.0 - b[i]) / (1.0 - ac[i]);
    }
    if (T < 2) {  // the reference indexes posterior_variance[1] (:150) and raises for T == 1
        gdmcf_set_error("schedule_build: steps must be >= 2");
        return GDMCF_E_SHAPE;
    }
    for (int i = 0; i < T; ++i) plv[i] = log(i == 0 ? pv[1] : pv[i]);
    return GDMCF_OK;
}

}  // extern "C"
