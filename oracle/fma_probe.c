/* fma_probe.c - test infrastructure (like everything under oracle/).
 *
 * The per-pixel 3x3 solve of GuidedFilter_cv (reference: src/CVF.cpp:116-147) restated as plain C, with NOTHING decided about
 * contraction: oracle/Makefile compiles this one file with `-O2 -mfma -ffp-contract=fast` (GCC's default contraction mode on an
 * FMA target - what a reference binary built for the ARM boards the project ran on contains), and tests/test_oracle.py compares
 * its output bit for bit with the oracle's reading PSMO_VAR_FMA_SOLVE (psm_oracle.c: psmo_solve_models) - the reading the
 * opt-in product form PSM_FLAG_FMA_SOLVE implements.  So "what GCC builds" is checked against a live GCC, not against our idea
 * of GCC.  x86-64 hosts with FMA only (the build skips it elsewhere, the test then skips too).
 */
#include <stddef.h>

#define GIF_EPS 0.0001f /* include/ComFunc.h:50 */

void psmo_probe_solve(const float *var_I, const float *cov, size_t N, float *a)
{
    for (size_t i = 0; i < N; ++i) {
        float c0 = cov[0 * N + i];
        float c1 = cov[1 * N + i];
        float c2 = cov[2 * N + i];
        float a11 = var_I[0 * N + i] + GIF_EPS;
        float a12 = var_I[1 * N + i];
        float a13 = var_I[2 * N + i];
        float a21 = var_I[1 * N + i];
        float a22 = var_I[3 * N + i] + GIF_EPS;
        float a23 = var_I[4 * N + i];
        float a31 = var_I[2 * N + i];
        float a32 = var_I[4 * N + i];
        float a33 = var_I[5 * N + i] + GIF_EPS;
        float DET = a11 * (a33 * a22 - a32 * a23) - a21 * (a33 * a12 - a32 * a13) +
                    a31 * (a23 * a12 - a22 * a13);
        DET = 1 / DET;
        a[0 * N + i] = DET * (c0 * (a33 * a22 - a32 * a23) + c1 * (a31 * a23 - a33 * a21) +
                              c2 * (a32 * a21 - a31 * a22));
        a[1 * N + i] = DET * (c0 * (a32 * a13 - a33 * a12) + c1 * (a33 * a11 - a31 * a13) +
                              c2 * (a31 * a12 - a32 * a11));
        a[2 * N + i] = DET * (c0 * (a23 * a12 - a22 * a13) + c1 * (a21 * a13 - a23 * a11) +
                              c2 * (a22 * a11 - a21 * a12));
    }
}
