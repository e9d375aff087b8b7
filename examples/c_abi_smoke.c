/* The C ABI from plain C (no Python, no C++):  the float64 schedule tables of the reference's default configuration
 * (models/gaussian_diffusion.py:109-159, linear-var, T=5) and the error convention.
 *   gcc -std=c99 examples/c_abi_smoke.c -Iinclude -Lgdmcf_amd/csrc -lgdmcf_hip -Wl,-rpath,$PWD/gdmcf_amd/csrc -o c_abi_smoke */
#include <math.h>
#include <stdio.h>

#include "gdmcf_hip.h"

int main(void) {
    enum { T = 5 };
    double tab[GDMCF_N_TABLES][T];
    int rc = gdmcf_schedule_build(1 /* linear-var */, 0.01, 0.001, 0.01, T, 1 /* beta_fixed */, &tab[0][0]);
    if (rc != GDMCF_OK) {
        fprintf(stderr, "schedule_build failed: %s\n", gdmcf_last_error());
        return 1;
    }
    /* pinned values of the reference */
    const double beta1 = 2.2500225002275442e-05, coef1_1 = 0.6923111538730923;
    if (fabs(tab[0][0] - 1e-5) > 0 || fabs(tab[0][1] - beta1) > 1e-18 || fabs(tab[11][1] - coef1_1) > 1e-13) {
        fprintf(stderr, "unexpected table values: %.17g %.17g %.17g\n", tab[0][0], tab[0][1], tab[11][1]);
        return 2;
    }
    rc = gdmcf_schedule_build(7 /* unknown schedule */, 0.01, 0.001, 0.01, T, 1, &tab[0][0]);
    printf("version %d; betas[1] = %.17g; posterior_mean_coef1[1] = %.17g; bad schedule -> rc %d (%s)\n", gdmcf_version(),
           tab[0][1], tab[11][1], rc, gdmcf_last_error());
    return rc == GDMCF_OK ? 3 : 0;
}
