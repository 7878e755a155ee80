// psm_exp.h - exp(x) for x <= 0 as the host's libm forms it: the algorithm of glibc >= 2.28 (sysdeps/ieee754/dbl-64/e_exp.c,
// from ARM's optimized-routines: exp(x) = 2^(k/N) exp(r), N = 128, x = k ln2/N + r, a table of 2^(k/N) split into H (1 + T) and
// a degree-5 polynomial) in the operation order of its FMA build (x86-64 with FMA, aarch64).  The weighted median's weights
// are (float)exp(double) (src/PP.cpp:175,224), and a median flips when a running sum sits within an ulp of half the total, so
// the weights have to be the host's bit for bit, not merely within an ulp of the true value as any libm's exp is.  This one is
// bit-for-bit glibc 2.35's on 5e7 arguments of the filter's domain (tests/test_oracle.py::test_wm_exp_is_the_host_libm_exp; its
// non-FMA build differs in the last bit of the DOUBLE for 0.07 % of them and in the float for none).  Plain C, shared by psm_pp.hip (PSM_EXP_FMA = __fma_rn) and the host check (fma);
// compile with -ffp-contract=off.
#ifndef PSM_EXP_H
#define PSM_EXP_H

#include "psm_exp_tab.h"

#ifndef PSM_EXP_FN
#define PSM_EXP_FN static inline
#endif

// x <= 0 (or NaN).  Bit-identical to glibc's exp for -150 < x <= 0; below that 0.0, which narrows to the same float (0.0f) as
// any double under 2^-150; NaN -> NaN.
PSM_EXP_FN double psm_exp_nonpos(double x, const unsigned long long *tab)
{
    const double InvLn2N = 0x1.71547652b82fep0 * 128, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47, Shift = 0x1.8p52;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    if (!(x > -150.0)) return x != x ? x : 0.0;
    if (x > -0x1p-54) return 1.0 + x;               // (0 is the common case: the window's centre)
    const double z = InvLn2N * x;
    double kd = z + Shift;
    union { double f; unsigned long long u; } cv;
    cv.f = kd;
    const unsigned long long ki = cv.u;
    kd -= Shift;
    const double r = PSM_EXP_FMA(kd, NegLn2loN, PSM_EXP_FMA(kd, NegLn2hiN, x));
    const unsigned long long idx = 2 * (ki % 128), top = ki << (52 - 7);
    cv.u = tab[idx];
    const double tail = cv.f;
    const unsigned long long sbits = tab[idx + 1] + top;
    const double r2 = r * r;
    const double t1 = tail + r;
    const double t2 = PSM_EXP_FMA(r2, PSM_EXP_FMA(r, C3, C2), t1);
    const double tmp = PSM_EXP_FMA(r2 * r2, PSM_EXP_FMA(r, C5, C4), t2);
    cv.u = sbits;
    const double scale = cv.f;
    return PSM_EXP_FMA(scale, tmp, scale);
}

#endif
