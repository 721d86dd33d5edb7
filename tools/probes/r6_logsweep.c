// Exhaustive CPU check of the bucket-table det_log_one_minus (rto_device_math.h, rto_log_table.h) against the oracle's
// orc_det_logf for every threshold draw u = k / 2^23: all operations are IEEE double (+ fma), which gfx950's v_fma_f64 /
// v_mul_f64 / v_add_f64 implement exactly, so a clean sweep here predicts a clean device sweep
// (tests/test_render_parity.py::test_every_threshold_draw_matches_the_oracle is the device's own).
//   gcc -O2 -ffp-contract=off -mfma -fopenmp -I rt-octree_amd/csrc -I oracle tools/probes/r6_logsweep.c oracle/rto_oracle.c -lm -o /tmp/logsweep
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#define __device__
#define namespace_rto
#include "rto_oracle.h"
// (the generated header is C++: strip its namespace / constexpr by hand)
#define constexpr static const
#define namespace struct
#undef namespace
#undef constexpr
#include "logtab_c.h"
static inline float fast_log_one_minus(float u01) {
    float a = 1.0f - u01;
    uint32_t u; memcpy(&u, &a, 4);
    int e = (int)(u >> 23) - 127;
    const uint32_t mf = u & 0x7fffffu;
    const int up = mf > 0x3504F3u;
    const uint32_t idx = ((mf + (1u << (22 - LOG2N))) >> (23 - LOG2N)) + (uint32_t)up;
    uint32_t mb = mf | 0x3f800000u; float mfl; memcpy(&mfl, &mb, 4);
    const double md = (double)mfl;
    const double r = fma(md, kLogTable[idx][0], -1.0);
    const double r2 = r * r;
    double p = -1.0 / 6.0;
    p = fma(p, r, 1.0 / 5.0);
    p = fma(p, r, -1.0 / 4.0);
    p = fma(p, r, 1.0 / 3.0);
    p = fma(p, r, -0.5);
    const double lm = fma(r2, p, r);
    const double t = fma((double)(e + up), 0.6931471805599453, kLogTable[idx][1]);
    return (float)(t + lm);
}
int main(void) {
    long long bad = 0;
#pragma omp parallel for reduction(+:bad)
    for (int64_t k = 0; k < (1 << 23); ++k) {
        uint32_t b = (uint32_t)k | 0x3f800000u; float f; memcpy(&f, &b, 4);
        const float u01 = f - 1.0f;
        const float ref = orc_det_logf(1.0f - u01), got = fast_log_one_minus(u01);
        uint32_t rb, gb; memcpy(&rb, &ref, 4); memcpy(&gb, &got, 4);
        if (rb != gb) {
            ++bad;
            if (bad < 20) printf("k %lld u %.9g ref %.9g (%08x) got %.9g (%08x)\n", (long long)k, u01, ref, rb, got, gb);
        }
    }
    printf("N = %d buckets: %lld of %d draws differ\n", 1 << LOG2N, bad, 1 << 23);
    return bad != 0;
}
