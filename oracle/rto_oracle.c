/*
 * rto_oracle.c -- CPU restatement of RT-Octree's batched-regular-tracking render path.
 * TEST INFRASTRUCTURE ONLY (see rto_oracle.h for the rules and the pinning status).
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Compile with -ffp-contract=off and no -ffast-math: the HIP kernels are compared against
 * this file bit for bit, so each expression keeps the reference's operand types (float vs
 * double sub-expressions are load-bearing) and its evaluation order.
 */
#include "rto_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_math_mode = ORC_MATH_DET;
void orc_set_math_mode(int mode) { g_math_mode = mode; }
int orc_get_math_mode(void) { return g_math_mode; }

/* render_options.hpp:15-58 defaults; opt.json overrides spp=6, denoise=true */
void orc_options_default(orc_options* o) {
    o->step_size = 1e-4f;
    o->sigma_thresh = 1e-2f;
    o->stop_thresh = 1e-2f;
    o->background_brightness = 1.f;
    o->render_bbox[0] = o->render_bbox[1] = o->render_bbox[2] = 0.f;
    o->render_bbox[3] = o->render_bbox[4] = o->render_bbox[5] = 1.f;
    o->basis_minmax[0] = 0;
    o->basis_minmax[1] = ORC_BASIS_MAX - 1;
    o->rot_dirs[0] = o->rot_dirs[1] = o->rot_dirs[2] = 0.f;
    o->denoise = 1;
    o->spp = 1;
}

/* ------------------------------------------------------------------ pcg32 */
#define PCG32_MULT 0x5851f42d4c957f2dULL /* pcg32.h:35 */

/* pcg32.h:62-68 */
uint32_t orc_pcg32_next_uint(orc_pcg32* r) {
    uint64_t oldstate = r->state;
    r->state = oldstate * PCG32_MULT + r->inc;
    uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
    uint32_t rot = (uint32_t)(oldstate >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

/* pcg32.h:53-59 */
void orc_pcg32_seed(orc_pcg32* r, uint64_t initstate, uint64_t initseq) {
    r->state = 0U;
    r->inc = (initseq << 1u) | 1u;
    orc_pcg32_next_uint(r);
    r->state += initstate;
    orc_pcg32_next_uint(r);
}

/* pcg32.h:103-112 */
float orc_pcg32_next_float(orc_pcg32* r) {
    union { uint32_t u; float f; } x;
    x.u = (orc_pcg32_next_uint(r) >> 9) | 0x3f800000u;
    return x.f - 1.0f;
}

/* pcg32.h:145-166 */
void orc_pcg32_advance(orc_pcg32* r, int64_t delta_) {
    uint64_t cur_mult = PCG32_MULT, cur_plus = r->inc, acc_mult = 1u, acc_plus = 0u;
    uint64_t delta = (uint64_t)delta_;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta /= 2;
    }
    r->state = acc_mult * r->state + acc_plus;
}

/* ------------------------------------------------------------------ math */
/* IEEE binary16 -> binary32, exact (the reference's __half2float, rt_core.cuh:251,290). */
float orc_half2float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    union { uint32_t u; float f; } v;
    if (exp == 0) {
        if (man == 0) {
            v.u = sign;
        } else { /* subnormal: man * 2^-24, exact in fp32 */
            float f = (float)man * 5.9604644775390625e-08f;
            v.f = f;
            v.u |= sign;
        }
    } else if (exp == 31) {
        v.u = sign | 0x7f800000u | (man << 13);
    } else {
        v.u = sign | ((exp + 112u) << 23) | (man << 13);
    }
    return v.f;
}

/*
 * Deterministic logf: the definition the HIP kernels share (rto_math.hip.h restates it with
 * the same constants).  x = m*2^e, m in (sqrt(1/2), sqrt(2)];  log m = 2 atanh((m-1)/(m+1)).
 * Only IEEE double + - * / and integer ops: identical bits on x86-64 and gfx950.
 */
float orc_det_logf(float x) {
    union { float f; uint32_t u; } v;
    v.f = x;
    if (x != x) return x;
    if (v.u == 0x7f800000u) return x;         /* +inf */
    if ((v.u << 1) == 0) return -INFINITY;    /* +-0 */
    if (v.u >> 31) return NAN;                /* negative */
    int e = (int)(v.u >> 23) - 127;
    uint32_t man = v.u & 0x7fffffu;
    if ((v.u >> 23) == 0) { /* subnormal: normalise */
        int sh = 0;
        while (!(man & 0x800000u)) { man <<= 1; ++sh; }
        man &= 0x7fffffu;
        e = -126 - sh;
    }
    union { uint32_t u; float f; } m;
    m.u = man | 0x3f800000u;
    double md = (double)m.f;
    if (md > 1.4142135623730951) { md = md * 0.5; e += 1; }
    double s = (md - 1.0) / (md + 1.0);
    double s2 = s * s;
    double p = 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2;
    double lm = 2.0 * s + (2.0 * s) * p;
    double r = (double)e * 0.6931471805599453 + lm;
    return (float)r;
}

/*
 * Deterministic expf: k = rint(x/ln2), r = x - k ln2 (two-term Cody-Waite), degree-11 Taylor
 * for e^r, scale by 2^k built from bits; one double->float rounding at the end.
 */
float orc_det_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -103.97208404541016f) return 0.0f;
    double xd = (double)x;
    double z = xd * 1.4426950408889634;
    double kd = (z + 6755399441055744.0) - 6755399441055744.0; /* 0x1.8p52: round to nearest */
    double r = (xd - kd * 0.693147180558298016) - kd * 1.6465949582897082e-12;
    double p = 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int k = (int)kd;
    union { uint64_t u; double d; } sc;
    sc.u = (uint64_t)(k + 1023) << 52;
    return (float)(p * sc.d);
}

/*
 * Deterministic fp32-only expf for the filter's softmax taps (164 per pixel at L = 4): the same
 * structure in float arithmetic, every multiply-add an explicit fmaf -- k = rint(x log2 e) by the
 * 1.5*2^23 trick, two-term Cody-Waite reduction, degree-6 minimax polynomial (Horner), exponent-bit scaling.
 * Max error 0.9 ulp on [-87.3, 88.7] (measured in tests/test_oracle_kat.py); below that the result is
 * a (coarser) subnormal, and exactly 0 from x = -87.68 down (the argument is clamped at -88, where k = -127
 * and the scale 2^k is written as +0).  The reference's __expf (filtering.cu:195) is ex2.approx(x*log2e), ~2 ulp plus the
 * rounding of x*log2e: this definition is at least as accurate.  fmaf is correctly rounded on both
 * sides (glibc / v_fma_f32), so the HIP kernel reproduces these bits.
 */
float orc_fexp(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    x = fmaxf(x, -88.0f); /* k = -127 there: the scale below is +0, i.e. exp flushes to 0 from -88 down */
    const float kf = fmaf(x, 1.44269502162933349609375f, 12582912.0f) - 12582912.0f;
    float r = fmaf(kf, -0.693145751953125f, x);
    r = fmaf(kf, -1.42860676533018704e-06f, r);
    float p = 0x1.6c6bdap-10f;      /* degree-6 minimax fit of exp on [-ln2/2, ln2/2] (tools/fit_fexp.py) */
    p = fmaf(p, r, 0x1.1225e0p-7f);
    p = fmaf(p, r, 0x1.5555a4p-5f);
    p = fmaf(p, r, 0x1.5554aep-3f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int ki = (int)kf; /* -127 .. 128 */
    if (ki > 127) { ki = 127; p = p * 2.0f; } /* 2^128 has no fp32 encoding */
    union { uint32_t u; float f; } sc;
    sc.u = (uint32_t)(ki + 127) << 23; /* ki = -127: +0.0 */
    return p * sc.f;
}

static inline float m_logf(float x) { return g_math_mode == ORC_MATH_DET ? orc_det_logf(x) : logf(x); }
static inline float m_expf(float x) { return g_math_mode == ORC_MATH_DET ? orc_det_expf(x) : expf(x); }
static inline float m_fexp(float x) { return g_math_mode == ORC_MATH_DET ? orc_fexp(x) : expf(x); }

/* CUDA min/max on floats (rt_core.cuh:33-34,48; common.hpp VOLREND_MIN/MAX): NaN-free here */
static inline float f_min(float a, float b) { return a < b ? a : b; }
static inline float f_max(float a, float b) { return a > b ? a : b; }

/* cuda/common.cuh:16-20 */
static inline float v_norm(const float* d) { return sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); }
/* cuda/common.cuh:22-27 */
static inline void v_normalize(float* d) {
    float invnorm = 1.f / v_norm(d);
    d[0] *= invnorm; d[1] *= invnorm; d[2] *= invnorm;
}

/* The deterministic math above on the floats with bit patterns first_bits + i * stride (wrapping): fn 0 = orc_det_logf,
 * 1 = orc_det_expf, 2 = orc_fexp.  The other half of the device-vs-oracle sweep (rto_probe_math). */
void orc_math_sweep(int fn, uint32_t first_bits, uint32_t stride, uint32_t count, float* out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)count; ++i) {
        union { uint32_t u; float f; } v;
        v.u = first_bits + (uint32_t)i * stride;
        out[i] = fn == 0 ? orc_det_logf(v.f) : fn == 1 ? orc_det_expf(v.f) : orc_fexp(v.f);
    }
}

/* rt_core.cuh:67-88: the value one threshold draw takes, t = -logf(1 - u), for the RNG float u = k / 2^23
 * (pcg32.h:103-112 next_float = ((next_uint >> 9) | 0x3f800000) - 1).  out[i] for k = first_k + i: lets a test
 * compare EVERY possible draw with the device function. */
void orc_thresholds(uint32_t first_k, uint32_t count, float* out) {
    for (uint32_t i = 0; i < count; ++i) {
        union { uint32_t u; float f; } v;
        v.u = ((first_k + i) & 0x7fffffu) | 0x3f800000u;
        out[i] = -m_logf(1.0f - (v.f - 1.0f));
    }
}

/* ------------------------------------------------------------------ sample_dst */
/* rt_core.cuh:67-193.  The 1..4 specialisations (:90-185) produce the same sorted array as the
 * generic insertion (:67-88); ties are indistinguishable in the result. */
void orc_sample_dst(int spp, orc_pcg32* rng, float* dst) {
    for (int n = 1; n <= spp; ++n) {
        float t = -m_logf(1.0f - orc_pcg32_next_float(rng));
        if (n == 1) {
            dst[0] = t;
        } else if (t <= dst[0]) {
            for (int i = n - 1; i > 0; i--) dst[i] = dst[i - 1];
            dst[0] = t;
        } else {
            int i = n - 1;
            while (dst[i - 1] > t) {
                dst[i] = dst[i - 1];
                i--;
            }
            dst[i] = t;
        }
    }
    dst[spp] = FLT_MAX; /* :192 */
}

/* ------------------------------------------------------------------ query */
/* internal/n3tree_query.hpp:13-48 */
int64_t orc_query(const orc_tree* t, float xyz[3], float* cube_sz, int* levels) {
    const float fN = (float)t->N;
    const int N3 = t->N * t->N * t->N;
    xyz[0] = f_max(f_min(xyz[0], 1.f - 1e-6f), 0.f);
    xyz[1] = f_max(f_min(xyz[1], 1.f - 1e-6f), 0.f);
    xyz[2] = f_max(f_min(xyz[2], 1.f - 1e-6f), 0.f);
    int64_t ptr = 0;
    *cube_sz = fN;
    int lv = 0;
    for (;;) {
        float index = 0.f;
        for (int i = 0; i < 3; ++i) {
            xyz[i] *= fN;
            const float idx_dimi = floorf(xyz[i]);
            index = index * fN + idx_dimi;
            xyz[i] -= idx_dimi;
        }
        const int64_t sub_ptr = ptr + (int32_t)index;
        const int64_t skip = t->child[sub_ptr];
        ++lv;
        if (skip == 0) {
            if (levels) *levels = lv;
            return sub_ptr;
        }
        *cube_sz *= fN;
        ptr += skip * N3;
    }
}

/* ------------------------------------------------------------------ SH basis */
/* internal/lumisphere.hpp:38-80.  The constants are double literals: each product is evaluated
 * in double and rounded once on assignment; integer-literal factors stay float. */
void orc_sh_basis(int basis_dim, const float dir[3], float out[ORC_BASIS_MAX]) {
    out[0] = 0.28209479177387814;
    const float x = dir[0], y = dir[1], z = dir[2];
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, yz = y * z, xz = x * z;
    switch (basis_dim) {
        case 25:
            out[16] = 2.5033429417967046 * xy * (xx - yy);
            out[17] = -1.7701307697799304 * yz * (3 * xx - yy);
            out[18] = 0.9461746957575601 * xy * (7 * zz - 1.f);
            out[19] = -0.6690465435572892 * yz * (7 * zz - 3.f);
            out[20] = 0.10578554691520431 * (zz * (35 * zz - 30) + 3);
            out[21] = -0.6690465435572892 * xz * (7 * zz - 3);
            out[22] = 0.47308734787878004 * (xx - yy) * (7 * zz - 1.f);
            out[23] = -1.7701307697799304 * xz * (xx - 3 * yy);
            out[24] = 0.6258357354491761 * (xx * (xx - 3 * yy) - yy * (3 * xx - yy));
            /* fallthrough */
        case 16:
            out[9] = -0.5900435899266435 * y * (3 * xx - yy);
            out[10] = 2.890611442640554 * xy * z;
            out[11] = -0.4570457994644658 * y * (4 * zz - xx - yy);
            out[12] = 0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy);
            out[13] = -0.4570457994644658 * x * (4 * zz - xx - yy);
            out[14] = 1.445305721320277 * z * (xx - yy);
            out[15] = -0.5900435899266435 * x * (xx - 3 * yy);
            /* fallthrough */
        case 9:
            out[4] = 1.0925484305920792 * xy;
            out[5] = -1.0925484305920792 * yz;
            out[6] = 0.31539156525252005 * (2.0 * zz - xx - yy);
            out[7] = -1.0925484305920792 * xz;
            out[8] = 0.5462742152960396 * (xx - yy);
            /* fallthrough */
        case 4:
            out[1] = -0.4886025119029199 * y;
            out[2] = 0.4886025119029199 * z;
            out[3] = -0.4886025119029199 * x;
    }
}

/* ------------------------------------------------------------------ trace_ray */
/* cuda/rt_core.cuh:195-332.  returns 0, or -1 for an unsupported format (SG/ASG: lumisphere.hpp
 * :14-37 "UNTESTED", out of scope per SURVEY section 2 #4). */
int orc_trace_ray(const orc_tree* tree, float dir[3], const float vdir[3], const float cen[3],
                  const orc_options* opt, float tmax_bg, float out[4], orc_pcg32* rng,
                  orc_stats* st) {
    const int SPP = opt->spp;
    /* _get_delta_scale :53-65 */
    dir[0] *= tree->scale[0];
    dir[1] *= tree->scale[1];
    dir[2] *= tree->scale[2];
    const float delta_scale = 1.f / v_norm(dir);
    dir[0] *= delta_scale;
    dir[1] *= delta_scale;
    dir[2] *= delta_scale;
    tmax_bg /= delta_scale; /* :208 */

    float tmin, tmax;
    float invdir[3];
    for (int i = 0; i < 3; ++i) invdir[i] = 1.f / (dir[i] + 1e-9); /* :214, double */
    /* _dda_world :19-36 */
    {
        float t1, t2;
        tmin = 0.0;
        tmax = 1e4;
        for (int i = 0; i < 3; ++i) {
            t1 = (opt->render_bbox[i] + 1e-6 - cen[i]) * invdir[i];      /* double */
            t2 = (opt->render_bbox[i + 3] - 1e-6 - cen[i]) * invdir[i];  /* double */
            tmin = f_max(tmin, f_min(t1, t2));
            tmax = f_min(tmax, f_max(t1, t2));
        }
    }
    tmax = f_min(tmax, tmax_bg); /* :217 */
    if (tmax < 0 || tmin > tmax) return 0; /* :219-222 */
    if (st) st->rays_in_box++;

    float pos[3], tmp;
    float t = tmin;
    float cube_sz;

    float src = 0;
    float dst[ORC_MAX_SPP + 1];
    orc_sample_dst(SPP, rng, dst); /* :232 */

    int64_t tree_vals[ORC_MAX_SPP];
    float cnts[ORC_MAX_SPP];
    memset(cnts, 0, sizeof(cnts));
    uint32_t spp = 0, sh_nums = 0;

    while (t < tmax) { /* :241-270 */
        pos[0] = cen[0] + t * dir[0];
        pos[1] = cen[1] + t * dir[1];
        pos[2] = cen[2] + t * dir[2];

        int lv;
        const int64_t leaf = orc_query(tree, pos, &cube_sz, &lv);
        if (st) { st->steps++; st->levels += (uint64_t)lv; }

        /* _dda_unit :38-51 */
        float t_unit;
        {
            float t1, t2;
            float tm = 1e4;
            for (int i = 0; i < 3; ++i) {
                t1 = -pos[i] * invdir[i];
                t2 = t1 + invdir[i];
                tm = f_min(tm, f_max(t1, t2));
            }
            t_unit = tm;
        }
        const float t_subcube = t_unit / cube_sz;
        const float delta_t = t_subcube + opt->step_size;
        const float sigma = orc_half2float(tree->data[leaf * tree->data_dim + tree->data_dim - 1]);
        if (sigma > opt->sigma_thresh) {
            const float delta = delta_t * delta_scale * sigma;
            if (src + delta >= dst[spp]) {
                float* cnt = &cnts[sh_nums];
                tree_vals[sh_nums] = leaf;
                ++sh_nums;
                do {
                    ++*cnt;
                    ++spp;
                } while (src + delta >= dst[spp]);
                if (spp == (uint32_t)SPP) break;
            }
            src += delta;
        }
        t += delta_t;
    }

    if (sh_nums == 0) return 0; /* :272-274 */
    if (st) { st->hit_rays++; st->hit_leaves += sh_nums; }

    const int basis_dim = tree->basis_dim;
    float basis_fn[ORC_BASIS_MAX];
    memset(basis_fn, 0, sizeof(basis_fn)); /* reference leaves k>=basis_dim uninitialised; never read */
    if (tree->format == ORC_FMT_SH) {
        orc_sh_basis(basis_dim, vdir, basis_fn); /* :278 */
    } else if (tree->format != ORC_FMT_RGBA) {
        return -1;
    }
    for (int i = 0; i < opt->basis_minmax[0] && i < ORC_BASIS_MAX; ++i) basis_fn[i] = 0.f; /* :279-281 */
    for (int i = opt->basis_minmax[1] + 1; i < ORC_BASIS_MAX; ++i)
        if (i >= 0) basis_fn[i] = 0.f; /* :282-284 */

    for (uint32_t i = 0; i < sh_nums; i++) { /* :286-325 */
        const uint16_t* tv = tree->data + tree_vals[i] * tree->data_dim;
        if (basis_dim >= 0) {
            int off = 0;
#define MUL_BASIS_I(k) (basis_fn[k] * orc_half2float(tv[off + (k)]))
            for (int c = 0; c < 3; ++c) {
                tmp = basis_fn[0] * orc_half2float(tv[off]);
                switch (basis_dim) {
                    case 25:
                        tmp += MUL_BASIS_I(16) + MUL_BASIS_I(17) + MUL_BASIS_I(18) + MUL_BASIS_I(19) +
                               MUL_BASIS_I(20) + MUL_BASIS_I(21) + MUL_BASIS_I(22) + MUL_BASIS_I(23) +
                               MUL_BASIS_I(24);
                        /* fallthrough */
                    case 16:
                        tmp += MUL_BASIS_I(9) + MUL_BASIS_I(10) + MUL_BASIS_I(11) + MUL_BASIS_I(12) +
                               MUL_BASIS_I(13) + MUL_BASIS_I(14) + MUL_BASIS_I(15);
                        /* fallthrough */
                    case 9:
                        tmp += MUL_BASIS_I(4) + MUL_BASIS_I(5) + MUL_BASIS_I(6) + MUL_BASIS_I(7) +
                               MUL_BASIS_I(8);
                        /* fallthrough */
                    case 4:
                        tmp += MUL_BASIS_I(1) + MUL_BASIS_I(2) + MUL_BASIS_I(3);
                }
                out[c] += cnts[i] / (1.f + m_expf(-tmp)); /* :314 */
                off += basis_dim;
            }
#undef MUL_BASIS_I
        } else {
            for (int j = 0; j < 3; ++j) out[j] += orc_half2float(tv[j]) * cnts[i]; /* :319-321 */
        }
        out[3] += cnts[i];
    }

    const float INV_SPP = 1.0f / (float)SPP; /* :327 */
    out[0] *= INV_SPP;
    out[1] *= INV_SPP;
    out[2] *= INV_SPP;
    out[3] *= INV_SPP;
    return 0;
}

/* ------------------------------------------------------------------ render_kernel */
static int spp_supported(int spp) { /* volrend.cu:266-278 */
    return spp == 1 || spp == 2 || spp == 3 || spp == 4 || spp == 6 || spp == 8 || spp == 16 || spp == 32;
}

/* src/cuda/volrend.cu:84-213, offscreen branch (ctx.offscreen = true, main_headless.cpp:450),
 * enable_probe = false. */
int orc_render_pixel(const orc_tree* tree, const orc_camera* cam, const orc_options* opt,
                     const orc_pcg32* rng_base, int idx, float aux8[8], float rgba[4], orc_stats* st) {
    if (!spp_supported(opt->spp)) return -2;
    const int x = idx % cam->width, y = idx / cam->width; /* :95 */
    float dir[3], cen[3], out[4];
    out[0] = out[1] = out[2] = out[3] = 0.f;
    const int enable_draw = tree->N > 0; /* :98 */
    float t_max = 1e9f;                  /* :136 */
    int rc = 0;
    if (enable_draw) {
        /* screen2worlddir :23-34 */
        float xyz[3] = {(x - 0.5f * cam->width) / cam->fx, -(y - 0.5f * cam->height) / cam->fy, -1.0f};
        const float* m = cam->transform;
        dir[0] = m[0] * xyz[0] + m[3] * xyz[1] + m[6] * xyz[2]; /* _mv3 common.cuh:29-37 */
        dir[1] = m[1] * xyz[0] + m[4] * xyz[1] + m[7] * xyz[2];
        dir[2] = m[2] * xyz[0] + m[5] * xyz[1] + m[8] * xyz[2];
        v_normalize(dir);
        cen[0] = m[9]; cen[1] = m[10]; cen[2] = m[11];
        float vdir[3] = {dir[0], dir[1], dir[2]}; /* :140 */
        /* maybe_world2ndc :35-56 */
        if (tree->ndc_width > 0) {
            float t = -(1.f + cen[2]) / dir[2];
            for (int i = 0; i < 3; ++i) cen[i] = cen[i] + t * dir[i];
            dir[0] = -((2 * tree->ndc_focal) / tree->ndc_width) * (dir[0] / dir[2] - cen[0] / cen[2]);
            dir[1] = -((2 * tree->ndc_focal) / tree->ndc_height) * (dir[1] / dir[2] - cen[1] / cen[2]);
            dir[2] = -2 / cen[2];
            cen[0] = -((2 * tree->ndc_focal) / tree->ndc_width) * (cen[0] / cen[2]);
            cen[1] = -((2 * tree->ndc_focal) / tree->ndc_height) * (cen[1] / cen[2]);
            cen[2] = 1 + 2 / cen[2];
            v_normalize(dir);
        }
        for (int i = 0; i < 3; ++i) cen[i] = tree->offset[i] + tree->scale[i] * cen[i]; /* :142-144 */
        /* rodrigues(opt.rot_dirs, vdir) :58-73,155 -- rotates the VIEW direction only (the SH lookup),
         * never the marching direction.  scalar_t = float: cos/sin resolve to the float overloads;
         * `(1.0 - cos_angle)` is a double, so the third term and the final sum are evaluated in
         * double and rounded once on assignment. */
        {
            float aa[3] = {opt->rot_dirs[0], opt->rot_dirs[1], opt->rot_dirs[2]};
            float angle = v_norm(aa);
            if (!(angle < 1e-6)) {
                float k[3];
                for (int i = 0; i < 3; ++i) k[i] = aa[i] / angle;
                float cos_angle = cosf(angle), sin_angle = sinf(angle);
                float cross[3];
                cross[0] = k[1] * vdir[2] - k[2] * vdir[1]; /* _cross3 common.cuh:54-60 */
                cross[1] = k[2] * vdir[0] - k[0] * vdir[2];
                cross[2] = k[0] * vdir[1] - k[1] * vdir[0];
                float dot = k[0] * vdir[0] + k[1] * vdir[1] + k[2] * vdir[2]; /* _dot3 :46-51 */
                for (int i = 0; i < 3; ++i)
                    vdir[i] = vdir[i] * cos_angle + cross[i] * sin_angle + k[i] * dot * (1.0 - cos_angle);
            }
        }
        orc_pcg32 rng = *rng_base;
        orc_pcg32_advance(&rng, (int64_t)(idx * opt->spp)); /* :157, int product */
        if (st) st->rays++;
        rc = orc_trace_ray(tree, dir, vdir, cen, opt, t_max, out, &rng, st);
        if (rc) return rc;
    }
    /* :174-179 */
    const float nalpha = 1.f - out[3];
    const float remain = opt->background_brightness * nalpha;
    out[0] += remain;
    out[1] += remain;
    out[2] += remain;
    /* :187-202 */
    aux8[0] = out[0]; aux8[1] = out[1]; aux8[2] = out[2]; aux8[3] = out[3];
    aux8[4] = out[0] * out[0]; aux8[5] = out[1] * out[1];
    aux8[6] = out[2] * out[2]; aux8[7] = out[3] * out[3];
    /* :205 */
    rgba[0] = out[0]; rgba[1] = out[1]; rgba[2] = out[2]; rgba[3] = 1.0f;
    return 0;
}

int orc_render_frame(const orc_tree* tree, const orc_camera* cam, const orc_options* opt,
                     const orc_pcg32* rng_base, float* aux, float* rgba, orc_stats* st,
                     int num_threads) {
    if (!spp_supported(opt->spp)) return -2;
    const int W = cam->width, H = cam->height;
    const int64_t SIZE = (int64_t)W * H;
    int err = 0;
    orc_stats total;
    memset(&total, 0, sizeof(total));
#ifdef _OPENMP
    if (num_threads <= 0) num_threads = omp_get_max_threads();
#else
    num_threads = 1;
#endif
#pragma omp parallel num_threads(num_threads)
    {
        orc_stats local;
        memset(&local, 0, sizeof(local));
#pragma omp for schedule(dynamic, 1)
        for (int y = 0; y < H; ++y) {
            for (int x = 0; x < W; ++x) {
                const int idx = y * W + x;
                float a8[8], px[4];
                int rc = orc_render_pixel(tree, cam, opt, rng_base, idx, a8, px, st ? &local : 0);
                if (rc) {
#pragma omp atomic write
                    err = rc;
                    continue;
                }
                for (int c = 0; c < 8; ++c) aux[c * SIZE + idx] = a8[c];
                for (int c = 0; c < 4; ++c) rgba[(int64_t)idx * 4 + c] = px[c];
            }
        }
#pragma omp critical
        {
            total.rays += local.rays; total.rays_in_box += local.rays_in_box;
            total.steps += local.steps; total.levels += local.levels;
            total.hit_leaves += local.hit_leaves; total.hit_rays += local.hit_rays;
        }
    }
    if (st) *st = total;
    return err;
}

/* ------------------------------------------------------------------ filter */
/* denoiser/extension/filtering.cu:108-228 (applying<_,16,32,SUPPORT>), driven level by level as
 * host::forward :440-470 does (support = level+1; level 0 overwrites with alpha=1, later levels
 * accumulate rgb).  Out-of-image taps: rgba=0, guidance=-FLT_MAX (:140-143). */
static int filter_core(int L, int H, int W, const float* weight, const float* guidance, const float* noisy,
                       float* out, float* rgb_filtered, float* max_map, float* inv_kernel_sum, int num_threads) {
    if (L < 1 || L > 6) return -4; /* kernel_apply :338-367 supports SUPPORT 1..6 */
#ifdef _OPENMP
    if (num_threads <= 0) num_threads = omp_get_max_threads();
    if (num_threads > (H + 3) / 4) num_threads = (H + 3) / 4; /* no more threads than chunks of rows */
#else
    num_threads = 1;
#endif
    for (int level = 0; level < L; ++level) {
        const int S = level + 1;
        const float* g = guidance + (int64_t)level * H * W;
        const float* wm = weight + (int64_t)level * H * W;
        /* (rows are dealt out dynamically, four at a time: with a static split of 800 rows over 256 threads one late thread
         *  holds up every level -- bench.py's cpu_baseline timed this leg at single-core speed on the 256-core box) */
#pragma omp parallel for schedule(dynamic, 4) num_threads(num_threads)
        for (int iy = 0; iy < H; ++iy) {
            for (int ix = 0; ix < W; ++ix) {
                float max_val = -FLT_MAX; /* :175-180 */
                for (int dy = -S; dy <= S; ++dy)
                    for (int dx = -S; dx <= S; ++dx) {
                        int qy = iy + dy, qx = ix + dx;
                        float kv = (qy >= 0 && qy < H && qx >= 0 && qx < W) ? g[(int64_t)qy * W + qx] : -FLT_MAX;
                        max_val = fmaxf(max_val, kv);
                    }
                float r = 0.f, gg = 0.f, b = 0.f, kernel_sum = 0; /* :183-199 */
                for (int dy = -S; dy <= S; ++dy)
                    for (int dx = -S; dx <= S; ++dx) {
                        int qy = iy + dy, qx = ix + dx;
                        int in = (qy >= 0 && qy < H && qx >= 0 && qx < W);
                        float kv = in ? g[(int64_t)qy * W + qx] : -FLT_MAX;
                        float k = m_fexp(kv - max_val);
                        kernel_sum += k;
                        const float* t = noisy + ((int64_t)qy * W + qx) * 4;
                        float tr = in ? t[0] : 0.f, tg = in ? t[1] : 0.f, tb = in ? t[2] : 0.f;
                        /* rgba.x += t_rgb.x * k (:197-199): nvcc contracts this to an FMA (the
                         * extension is built with nvcc's defaults, network.py:16-18: -fmad=true);
                         * stated explicitly so CPU and GPU agree */
                        r = fmaf(tr, k, r);
                        gg = fmaf(tg, k, gg);
                        b = fmaf(tb, k, b);
                    }
                const float inv = 1.0f / kernel_sum;                 /* :204 */
                if (rgb_filtered) {                                  /* saved for backward :205-216 */
                    const int64_t si = ((int64_t)level * H + iy) * W + ix;
                    max_map[si] = max_val;
                    inv_kernel_sum[si] = inv;
                    rgb_filtered[si * 4 + 0] = r * inv;
                    rgb_filtered[si * 4 + 1] = gg * inv;
                    rgb_filtered[si * 4 + 2] = b * inv;
                    rgb_filtered[si * 4 + 3] = 0.f; /* torch::zeros, never written (:622) */
                }
                const float w = wm[(int64_t)iy * W + ix] * inv;      /* :218 */
                r *= w; gg *= w; b *= w;
                float* o = out + ((int64_t)iy * W + ix) * 4;
                if (S == 1) { o[0] = r; o[1] = gg; o[2] = b; o[3] = 1.0f; } /* :47-60 */
                else { o[0] += r; o[1] += gg; o[2] += b; }                  /* :62-74 */
            }
        }
    }
    return 0;
}

int orc_filter(int L, int H, int W, const float* weight, const float* guidance, const float* noisy,
               float* out, int num_threads) {
    return filter_core(L, H, W, weight, guidance, noisy, out, NULL, NULL, NULL, num_threads);
}

/* Filtering::forward with requires_grad (filtering.cu:596-665): the same pass, also saving per level
 * rgb_filtered [L][H][W][4], max_map [L][H][W], inv_kernel_sum [L][H][W] (the reference keeps one
 * tensor per level, [B,H,W,4] / [B,H,W]; here one image, levels outermost). */
int orc_filter_train_forward(int L, int H, int W, const float* weight, const float* guidance, const float* noisy,
                             float* out, float* rgb_filtered, float* max_map, float* inv_kernel_sum, int num_threads) {
    if (!rgb_filtered || !max_map || !inv_kernel_sum) return -1;
    return filter_core(L, H, W, weight, guidance, noisy, out, rgb_filtered, max_map, inv_kernel_sum, num_threads);
}

/* Filtering::backward (filtering.cu:667-707) = per level grad_weight_accumulate (:230-248) and
 * grad_guidance_accumulate (:250-301):
 *   grad_weight[l][p]   = sum_c grad_out[p][c] * rgb_filtered_l[p][c]
 *   grad_guidance[l][q] = sum over the pixels p whose (2S+1)^2 window holds q of
 *                         w_l[p] * (exp(g_l[q] - max_l[p]) * inv_l[p]) * sum_c grad_out[p][c] * (img_in[q][c] - rgb_filtered_l[p][c])
 * The reference scatters the second sum with one thread per (p, tap) and atomicAdd, i.e. in an order
 * the hardware picks; this restatement gathers per q with p in row-major window order, which fixes the
 * result.  Products-into-sums are explicit fmaf (nvcc's default contraction, as in the forward). */
int orc_filter_backward(int L, int H, int W, const float* grad_out, const float* img_in, const float* weight,
                        const float* guidance, const float* rgb_filtered, const float* max_map,
                        const float* inv_kernel_sum, float* grad_weight, float* grad_guidance, int num_threads) {
    if (L < 1 || L > 6) return -4;
#ifdef _OPENMP
    if (num_threads <= 0) num_threads = omp_get_max_threads();
#else
    num_threads = 1;
#endif
    for (int level = 0; level < L; ++level) {
        const int S = level + 1;
        const int64_t lo = (int64_t)level * H * W;
#pragma omp parallel for schedule(static) num_threads(num_threads)
        for (int qy = 0; qy < H; ++qy) {
            for (int qx = 0; qx < W; ++qx) {
                const int64_t q = (int64_t)qy * W + qx;
                { /* grad_weight :244-247 */
                    const float* go = grad_out + q * 4;
                    const float* f = rgb_filtered + (lo + q) * 4;
                    float t = go[0] * f[0];
                    t = fmaf(go[1], f[1], t);
                    t = fmaf(go[2], f[2], t);
                    grad_weight[lo + q] = t;
                }
                const float gq = guidance[lo + q];
                const float* in = img_in + q * 4;
                float acc = 0.f;
                for (int py = qy - S; py <= qy + S; ++py)
                    for (int px = qx - S; px <= qx + S; ++px) {
                        if (py < 0 || py >= H || px < 0 || px >= W) continue; /* threads exist for image pixels only */
                        const int64_t pi = (int64_t)py * W + px;
                        const float* go = grad_out + pi * 4;
                        const float* f = rgb_filtered + (lo + pi) * 4;
                        const float k = m_fexp(gq - max_map[lo + pi]) * inv_kernel_sum[lo + pi]; /* :293 */
                        float res = go[0] * (in[0] - f[0]);                                     /* :294-297 */
                        res = fmaf(go[1], in[1] - f[1], res);
                        res = fmaf(go[2], in[2] - f[2], res);
                        res *= weight[lo + pi] * k;                                              /* :298 */
                        acc += res;                                                              /* :300 */
                    }
                grad_guidance[lo + q] = acc;
            }
        }
    }
    return 0;
}

/* main_headless.cpp:535-538: buf_uint8[j] = buf[j] * 255 (float -> uint8 truncation) */
void orc_rgba8(const float* rgba, uint8_t* out, int64_t n) {
    for (int64_t j = 0; j < n; ++j) out[j] = (uint8_t)(rgba[j] * 255);
}

/* Instrumentation only: march steps per pixel of one frame (how unevenly the work is spread). */
int orc_frame_steps(const orc_tree* tree, const orc_camera* cam, const orc_options* opt,
                    const orc_pcg32* rng_base, uint32_t* steps_out, int num_threads) {
    const int W = cam->width, H = cam->height;
    int err = 0;
#ifdef _OPENMP
    if (num_threads <= 0) num_threads = omp_get_max_threads();
#else
    num_threads = 1;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            orc_stats st;
            memset(&st, 0, sizeof(st));
            float a8[8], px[4];
            int rc = orc_render_pixel(tree, cam, opt, rng_base, y * W + x, a8, px, &st);
            if (rc) err = rc;
            steps_out[y * W + x] = (uint32_t)st.steps;
        }
    return err;
}
