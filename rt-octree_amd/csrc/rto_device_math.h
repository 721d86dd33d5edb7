// rto_device_math.h -- device-side scalar helpers shared by the gfx950 kernels.
//
// The arithmetic here is the *definition* the CPU oracle checks bit for bit, so:
//   * no FMA contraction (file-level pragma below + -ffp-contract=off on the command line),
//   * logf/expf are built from IEEE double + - * / only (the reference's __logf/__expf,
//     rt_core.cuh:74,95,314 and filtering.cu:191, are NVIDIA approximations that cannot be
//     reproduced; DESIGN.md "Math"),
//   * fp32 divide / sqrt stay correctly rounded (hipcc default).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace rto {

#define RTO_DEV __device__ __forceinline__

// ---- pcg32 (renderer/3rdparty/pcg32.h:39-201) ----
struct Pcg32 {
    uint64_t state, inc;
};
constexpr uint64_t kPcgMult = 0x5851f42d4c957f2dULL;  // pcg32.h:35

RTO_DEV uint32_t pcg_next_uint(Pcg32& r) {  // pcg32.h:62-68
    const uint64_t oldstate = r.state;
    r.state = oldstate * kPcgMult + r.inc;
    const uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
    const uint32_t rot = (uint32_t)(oldstate >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

RTO_DEV float pcg_next_float(Pcg32& r) {  // pcg32.h:103-112
    return __uint_as_float((pcg_next_uint(r) >> 9) | 0x3f800000u) - 1.0f;
}

// pcg32.h:145-166, Brown's O(log n) jump
RTO_DEV void pcg_advance(Pcg32& r, int64_t delta_) {
    uint64_t cur_mult = kPcgMult, cur_plus = r.inc, acc_mult = 1u, acc_plus = 0u;
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
    r.state = acc_mult * r.state + acc_plus;
}

// Same jump through four 256-entry tables of (mult, plus) for delta = j << (8*c): the affine
// maps commute, so applying the four byte-chunks in any order lands on the identical state
// (arithmetic mod 2^64 is exact).  Tables are built on the host per `inc` (rto_abi.hip).
struct PcgJumpEntry {
    uint64_t mult, plus;
};
RTO_DEV void pcg_advance_tab(Pcg32& r, uint32_t delta, const PcgJumpEntry* __restrict__ tab) {
    uint64_t s = r.state;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // (entry 0 of every digit is the identity: a digit that is zero in all lanes of the wave -- the top byte of
        //  idx * SPP for any frame below 2^24 samples -- costs one ballot instead of a gather and a 64-bit multiply)
        const uint32_t j = (delta >> (8 * c)) & 255u;
        if (c >= 2 && __ballot(j != 0u) == 0ULL) continue;
        const PcgJumpEntry e = tab[c * 256 + j];
        s = e.mult * s + e.plus;
    }
    r.state = s;
}

// ---- fp16 -> fp32 (exact) ----
RTO_DEV float half_bits_to_float(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

// (float)h * b for the fp16 value h in the low / high half of `packed`, in ONE instruction: v_fma_mix_f32 widens the half
// operand itself, and a * b + (-0.0) is a * b for every a, b (same rounding, same sign of zero, same infinities) -- the
// conversion v_cvt_f32_f16 is exact, so this is the float of half_bits_to_float(h) * b.  The shading kernels multiply 48
// coefficients per hit entry (rt_core.cuh:286-312): 48 of ~530 vector instructions per 64 entries saved (SQ_INSTS_VALU -8 %); the
// kernel's time did not move (1.333 -> 1.330 ms per 100 C2 frames, profiles/r6_t_ab_shade_mix.txt): v_fma_mix_f32 issues at half
// rate like the conversion it replaces (profiles/r6_v_probe_f64.txt), and the kernel is not bound by issue.  Compared on the device with
// the two-instruction form for every half (denormals, infinities) against 2^20 floats and for 16 special halves against all 2^32
// floats (rto_probe_sigmoid modes 2 / 3, tests/test_render_parity.py).
RTO_DEV float mul_half_lo(uint32_t packed, float b) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, neg(0) op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(b));
    return r;
}
RTO_DEV float mul_half_hi(uint32_t packed, float b) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, neg(0) op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(b));
    return r;
}

// ---- deterministic logf / expf: constants and operation order identical to
//      oracle/rto_oracle.c orc_det_logf / orc_det_expf ----
// the reduction + series of det_logf for a POSITIVE NORMAL finite x (exponent e, mantissa bits man)
RTO_DEV float det_logf_core(int e, uint32_t man) {
    double md = (double)__uint_as_float(man | 0x3f800000u);
    if (md > 1.4142135623730951) {
        md = md * 0.5;
        e += 1;
    }
    const double s = (md - 1.0) / (md + 1.0);
    const double s2 = s * s;
    double p = 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2;
    const double lm = 2.0 * s + (2.0 * s) * p;
    const double r = (double)e * 0.6931471805599453 + lm;
    return (float)r;
}

RTO_DEV float det_logf(float x) {
    const uint32_t u = __float_as_uint(x);
    if (x != x) return x;
    if (u == 0x7f800000u) return x;
    if ((u << 1) == 0) return -__builtin_inff();
    if (u >> 31) return __builtin_nanf("");
    int e = (int)(u >> 23) - 127;
    uint32_t man = u & 0x7fffffu;
    if ((u >> 23) == 0) {
        int sh = 0;
        while (!(man & 0x800000u)) {
            man <<= 1;
            ++sh;
        }
        man &= 0x7fffffu;
        e = -126 - sh;
    }
    return det_logf_core(e, man);
}

// det_logf(1 - u) for u = pcg_next_float() = k / 2^23, k < 2^23: the argument is a positive normal number in
// [2^-23, 1], so none of det_logf's special cases (NaN, infinity, zero, negative, subnormal) can occur.  The
// threshold draws (SPP per pixel; sample_kernel is VALU-bound) evaluate the same reduction and series with the
// quotient from v_rcp_f64 + two Newton steps + one residual correction instead of the IEEE division sequence (~25
// instructions) and with fused multiply-adds in the polynomial: the intermediate doubles differ from det_logf_core's in
// the last bit or two, the float they round to does not -- for ALL 2^23 possible draws, which
// tests/test_render_parity.py::test_every_threshold_draw_matches_the_oracle checks one by one against
// oracle/rto_oracle.c (rto_probe_thresholds).
RTO_DEV float det_log_one_minus(float u01) {
    const uint32_t u = __float_as_uint(1.0f - u01);
    int e = (int)(u >> 23) - 127;
    double md = (double)__uint_as_float((u & 0x7fffffu) | 0x3f800000u);
    if (md > 1.4142135623730951) {
        md = md * 0.5;
        e += 1;
    }
    const double n = md - 1.0, d = md + 1.0;  // exact: md has 24 significant bits
    double y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    double s = n * y;
    s = __builtin_fma(__builtin_fma(-d, s, n), y, s);
    const double s2 = s * s;
    double p = 1.0 / 15.0;
    p = __builtin_fma(p, s2, 1.0 / 13.0);
    p = __builtin_fma(p, s2, 1.0 / 11.0);
    p = __builtin_fma(p, s2, 1.0 / 9.0);
    p = __builtin_fma(p, s2, 1.0 / 7.0);
    p = __builtin_fma(p, s2, 1.0 / 5.0);
    p = __builtin_fma(p, s2, 1.0 / 3.0);
    p = p * s2;
    const double s_2 = 2.0 * s;
    const double lm = __builtin_fma(s_2, p, s_2);
    return (float)__builtin_fma((double)e, 0.6931471805599453, lm);
}

RTO_DEV float det_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return __builtin_inff();
    if (x < -103.97208404541016f) return 0.0f;
    const double xd = (double)x;
    const double z = xd * 1.4426950408889634;
    const double kd = (z + 6755399441055744.0) - 6755399441055744.0;
    const double r = (xd - kd * 0.693147180558298016) - kd * 1.6465949582897082e-12;
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
    const int k = (int)kd;
    const double sc = __longlong_as_double((long long)((uint64_t)(k + 1023) << 52));
    return (float)(p * sc);
}

// det_expf for a finite argument |x| <= 87 (no special case can occur: the result is a normal float), in 21 vector
// instructions instead of ~55 + three nested branches (round 6; shading spends three of these per hit entry): the same
// reduction -- z, k = rint(z) as v_rndne_f64 (what the 1.5 * 2^52 add / subtract computes for |z| < 2^51), the two
// un-fused Cody-Waite steps, so r is det_expf's r bit for bit -- then the same degree-11 polynomial as FUSED multiply-adds
// and the scale 2^k as v_ldexp_f64 (a power of two commutes with the rounding).  The double differs from det_expf's by at
// most ONE unit in the last place, never across a float rounding boundary: checked for every one of the 2 237 399 042 floats
// with |x| <= 87 on the CPU (tools/probes/r6_expsweep.c: all operations are IEEE double, which v_fma_f64 / v_rndne_f64 /
// v_ldexp_f64 implement exactly) and on the device against det_expf itself (tests: rto_probe_sigmoid, all 2^32 floats).
RTO_DEV float det_expf_mid(float x) {
    const double xd = (double)x;
    const double kd = __builtin_rint(xd * 1.4426950408889634);
    const double r = (xd - kd * 0.693147180558298016) - kd * 1.6465949582897082e-12;
    double p = 1.0 / 39916800.0;
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return (float)__builtin_ldexp(p, (int)kd);
}

// cnt / d for d in [1, 2^126) and cnt = 1 .. 32 (a hit entry's sample count over 1 + e^-t): v_rcp_f32 (1 ulp), the quotient
// estimate, its exact residual (one fma) and one correction -- 4 instructions for the ~11 of the IEEE division sequence
// (v_div_scale x 2, v_rcp, 4 fma, v_div_fmas, v_div_fixup), which guards exponent ranges these operands cannot reach.
// The same float as the division for EVERY such d and cnt (126 * 2^23 * 32 cases, rto_probe_sigmoid on the device:
// v_rcp_f32's bits are the hardware's, so only the device can check it).
RTO_DEV float div_small_by_ge1(float cnt, float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = cnt * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, cnt), r, q);
}

// rt_core.cuh:314-318 for the three colour channels of one hit leaf: o[c] = cnt / (1 + exp(-t[c])).  Every lane whose three
// arguments lie in [-87, 87] -- every lane, on any scene whose SH sums are not absurd -- takes the short forms above; a wave
// with a lane outside (NaN, infinities, |t| > 87: overflow, underflow and subnormal results) sends that lane through the
// plain statement.  Same floats either way.
RTO_DEV void sigmoid_cnt3(const float* t, float cnt, float* o) {
#ifdef RTO_SIGMOID_PLAIN  // (same-box A/B of round 6 only)
    for (int c = 0; c < 3; ++c) o[c] = cnt / (1.f + det_expf(-t[c]));
    return;
#endif
    const bool plain = !(__builtin_fabsf(t[0]) <= 87.f) || !(__builtin_fabsf(t[1]) <= 87.f) || !(__builtin_fabsf(t[2]) <= 87.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = div_small_by_ge1(cnt, 1.f + det_expf_mid(-t[c]));
    if (__builtin_expect(__ballot(plain) != 0ULL, 0)) {
        if (plain) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = cnt / (1.f + det_expf(-t[c]));
        }
    }
}

// fp32-only deterministic exp for the filter taps; mirrors oracle/rto_oracle.c orc_fexp (every
// multiply-add an explicit, correctly rounded fma: v_fma_f32 here, fmaf there)
RTO_DEV float fexp_f32(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return __builtin_inff();
    x = __builtin_fmaxf(x, -88.0f);
    const float kf = __builtin_fmaf(x, 1.44269502162933349609375f, 12582912.0f) - 12582912.0f;
    float r = __builtin_fmaf(kf, -0.693145751953125f, x);
    r = __builtin_fmaf(kf, -1.42860676533018704e-06f, r);
    float p = 0x1.6c6bdap-10f;  // degree-6 minimax fit (tools/fit_fexp.py), <= 0.92 ulp overall
    p = __builtin_fmaf(p, r, 0x1.1225e0p-7f);
    p = __builtin_fmaf(p, r, 0x1.5555a4p-5f);
    p = __builtin_fmaf(p, r, 0x1.5554aep-3f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    int ki = (int)kf;
    if (ki > 127) {  // 2^128 has no fp32 encoding
        ki = 127;
        p = p * 2.0f;
    }
    return p * __uint_as_float((uint32_t)(ki + 127) << 23);
}

// The same function without control flow, for finite arguments x <= 88 (the filter only ever passes
// g - max <= 0): identical values.  The scale 2^k comes straight from the bits of the rounding
// constant trick: t = x*log2e + 1.5*2^23 holds k in its low mantissa bits (0x4B400000 + k as an
// integer), and (t_bits << 23) + 0x3f800000 keeps exactly (k + 127) << 23 -- no float->int convert.
RTO_DEV float fexp_f32_le88(float x) {
    x = __builtin_fmaxf(x, -88.0f);
    const float t = __builtin_fmaf(x, 1.44269502162933349609375f, 12582912.0f);
    const float kf = t - 12582912.0f;
    float r = __builtin_fmaf(kf, -0.693145751953125f, x);
    r = __builtin_fmaf(kf, -1.42860676533018704e-06f, r);
    float p = 0x1.6c6bdap-10f;  // degree-6 minimax fit (tools/fit_fexp.py), <= 0.92 ulp overall
    p = __builtin_fmaf(p, r, 0x1.1225e0p-7f);
    p = __builtin_fmaf(p, r, 0x1.5555a4p-5f);
    p = __builtin_fmaf(p, r, 0x1.5554aep-3f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    return p * __uint_as_float((__float_as_uint(t) << 23) + 0x3f800000u);
}

// Two arguments at once: the same operations on a 2-vector, which gfx950 executes as packed fp32
// instructions (v_pk_fma_f32 / v_pk_add_f32: two IEEE results per lane per instruction, each
// rounded exactly like the scalar form).
typedef float float2v __attribute__((ext_vector_type(2)));
typedef int int2v __attribute__((ext_vector_type(2)));
RTO_DEV float2v splat2(float v) { return float2v{v, v}; }
RTO_DEV float2v fexp_f32_le88_x2(float2v x) {
    x.x = __builtin_fmaxf(x.x, -88.0f);
    x.y = __builtin_fmaxf(x.y, -88.0f);
    const float2v t = __builtin_elementwise_fma(x, splat2(1.44269502162933349609375f), splat2(12582912.0f));
    const float2v kf = t - 12582912.0f;
    float2v r = __builtin_elementwise_fma(kf, splat2(-0.693145751953125f), x);
    r = __builtin_elementwise_fma(kf, splat2(-1.42860676533018704e-06f), r);
    float2v p = splat2(0x1.6c6bdap-10f);
    p = __builtin_elementwise_fma(p, r, splat2(0x1.1225e0p-7f));
    p = __builtin_elementwise_fma(p, r, splat2(0x1.5555a4p-5f));
    p = __builtin_elementwise_fma(p, r, splat2(0x1.5554aep-3f));
    p = __builtin_elementwise_fma(p, r, splat2(0.5f));
    p = __builtin_elementwise_fma(p, r, splat2(1.0f));
    p = __builtin_elementwise_fma(p, r, splat2(1.0f));
    float2v sc;
    sc.x = __uint_as_float((__float_as_uint(t.x) << 23) + 0x3f800000u);
    sc.y = __uint_as_float((__float_as_uint(t.y) << 23) + 0x3f800000u);
    return p * sc;
}

// 1 / x by v_rcp_f32 (1 ulp) and one Newton step: within an ulp of the IEEE quotient in 3 instructions instead of 11.  For the
// tolerance routes only (the factorised filter, the softmax of the network's logits); x normal, not 0.
RTO_DEV float rcp_refined(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
// n / d as n * rcp(d) with one exact-residual correction: the correctly rounded quotient except for rare last-bit cases, in 4
// instructions (v_rcp_f32, multiply, two FMAs) -- for the tolerance routes; d normal, not 0, the quotient far from the range's ends.
RTO_DEV float div_refined(float n, float d) {
    const float r = __builtin_amdgcn_rcpf(d), q = n * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}

// weight = softmax(x[:, :4]) (network.py:113-114) in fp32 over the four fp16-valued logits; one definition for
// the GuidanceNet kernel's epilogue and for the filter that consumes packed logits, so both give the same bits
RTO_DEV void softmax_weights4(const float* v, float* out) {
    const float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    float e[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        e[i] = __expf(v[i] - m);
        s += e[i];
    }
    const float inv = rcp_refined(s);  // s in [1, 4]
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = e[i] * inv;
}

RTO_DEV float f_min(float a, float b) { return a < b ? a : b; }
RTO_DEV float f_max(float a, float b) { return a > b ? a : b; }

}  // namespace rto
