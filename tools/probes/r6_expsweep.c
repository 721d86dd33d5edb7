// exhaustive CPU check of the short exp core vs the oracle's det_expf (all IEEE double ops: identical on gfx950)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
static inline double slow_d(float x) {
    double xd = (double)x;
    double z = xd * 1.4426950408889634;
    double kd = (z + 6755399441055744.0) - 6755399441055744.0;
    double r = (xd - kd * 0.693147180558298016) - kd * 1.6465949582897082e-12;
    double p = 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0; p = p * r + 1.0 / 362880.0; p = p * r + 1.0 / 40320.0; p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0; p = p * r + 1.0 / 120.0; p = p * r + 1.0 / 24.0; p = p * r + 1.0 / 6.0;
    p = p * r + 0.5; p = p * r + 1.0; p = p * r + 1.0;
    int k = (int)kd;
    union { uint64_t u; double d; } sc; sc.u = (uint64_t)(k + 1023) << 52;
    return p * sc.d;
}
static inline double fast_d(float x, int variant) {
    double xd = (double)x;
    double kd = rint(xd * 1.4426950408889634);
    double r;
    if (variant == 0) { r = fma(-kd, 0.693147180558298016, xd); r = fma(-kd, 1.6465949582897082e-12, r); }
    else r = (xd - kd * 0.693147180558298016) - kd * 1.6465949582897082e-12;
    double p = 1.0 / 39916800.0;
    p = fma(p, r, 1.0 / 3628800.0); p = fma(p, r, 1.0 / 362880.0); p = fma(p, r, 1.0 / 40320.0); p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0); p = fma(p, r, 1.0 / 120.0); p = fma(p, r, 1.0 / 24.0); p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5); p = fma(p, r, 1.0); p = fma(p, r, 1.0);
    return ldexp(p, (int)kd);
}
int main(int argc, char** argv) {
    int variant = argc > 1 ? atoi(argv[1]) : 0;
    uint32_t lim; float f87 = 87.0f; memcpy(&lim, &f87, 4);
    long long raw = 0, n = 0, maxulp = 0, uncaught[6] = {0}, flagged[6] = {0};
    const long long M[6] = {8, 32, 64, 128, 256, 512};
#pragma omp parallel for reduction(+:raw,n) schedule(dynamic, 1 << 20)
    for (int64_t b = 0; b <= 2 * (int64_t)lim + 1; ++b) {
        uint32_t bits = b <= lim ? (uint32_t)b : 0x80000000u | (uint32_t)(b - lim - 1);
        float x; memcpy(&x, &bits, 4);
        double s = slow_d(x), f = fast_d(x, variant);
        int64_t sb, fb; memcpy(&sb, &s, 8); memcpy(&fb, &f, 8);
        int64_t d = llabs(sb - fb);
        int mism = (float)s != (float)f;
        raw += mism; ++n;
        uint32_t lo = (uint32_t)fb;
        if (d > maxulp || mism) {
#pragma omp critical
            {
                if (d > maxulp) maxulp = d;
                if (mism) for (int i = 0; i < 6; ++i) {
                    uint32_t t = (lo - (0x10000000u - (uint32_t)M[i])) & 0x1fffffffu;
                    if (!(t <= 2 * M[i])) ++uncaught[i];
                }
            }
        }
        if ((b & 0xfff) == 0) {  // sample the flag rate
            for (int i = 0; i < 6; ++i) { uint32_t t = (lo - (0x10000000u - (uint32_t)M[i])) & 0x1fffffffu; if (t <= 2 * M[i]) {
#pragma omp atomic
                ++flagged[i]; } }
        }
    }
    printf("variant %d: inputs %lld raw float mismatches %lld max double ulp distance %lld\n", variant, n, raw, maxulp);
    for (int i = 0; i < 6; ++i) printf("  margin %lld: uncaught %lld, flagged (of %lld sampled) %lld\n", M[i], uncaught[i], n >> 12, flagged[i]);
    return 0;
}
