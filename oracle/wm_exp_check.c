/* Checker (test infrastructure): the product's exp for non-positive arguments (primestereomatch_amd/csrc/psm_exp.h, the
 * weighted median's weights) against THIS host's libm exp, on the filter's own argument domain and on uniformly random
 * arguments.  Prints "<n> <double mismatches> <float mismatches>"; exit status 0 when both are 0.
 *   gcc -O2 -ffp-contract=off -o wm_exp_check wm_exp_check.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define PSM_EXP_FMA(a, b, c) fma(a, b, c)
#include "../primestereomatch_amd/csrc/psm_exp.h"

static const unsigned long long tab[256] = PSM_EXP_TAB_INIT;

static uint64_t bits(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    long bad = 0, badf = 0;
    uint64_t s = 88172645463325252ull;
    for (long i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double x;
        switch (i & 3) {
        case 0: x = -(double)(s >> 11) * (1.0 / 9007199254740992.0) * 160.0; break;   /* uniform in (-160, 0] */
        case 1: {   /* src/PP.cpp:169-175: -disWgt / 81 (float) - clrWgt / 0.01 (double), squared distances */
            const float c = (float)((s >> 20) & 0xffffff) * (3.0f / 16777216.0f), d = (float)((s >> 8) % 163);
            x = (double)(-d / 81.0f) - (double)c / (0.1 * 0.1);
            break;
        }
        case 2: {   /* :216-224: the square-rooted ones */
            const float c = sqrtf((float)((s >> 20) & 0xffffff) * (3.0f / 16777216.0f)), d = sqrtf((float)((s >> 8) % 163));
            x = (double)(-d / 81.0f) - (double)c / (0.1 * 0.1);
            break;
        }
        default: x = -ldexp((double)(s >> 11), -53 - (int)(s & 63)); break;           /* small magnitudes down to 2^-117 */
        }
        const double a = exp(x), b = psm_exp_nonpos(x, tab);
        if (x > -150.0 && bits(a) != bits(b)) ++bad;
        if ((float)a != (float)b) ++badf;
    }
    const double edge[] = {0.0, -0.0, -0x1p-60, -149.99, -150.0, -151.0, -745.0, -1e9, -INFINITY};
    for (unsigned i = 0; i < sizeof edge / sizeof edge[0]; ++i)
        if ((float)exp(edge[i]) != (float)psm_exp_nonpos(edge[i], tab)) ++badf;
    if (!isnan(psm_exp_nonpos(NAN, tab))) ++badf;
    printf("%ld %ld %ld\n", n, bad, badf);
    return bad || badf;
}
