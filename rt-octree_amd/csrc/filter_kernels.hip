// filter_kernels.hip -- multi-level guided softmax filter (the GuidanceNet "kernel applying" stage).
//
// Reference: denoiser/extension/filtering.cu:108-228 `kernel::applying<Out,16,32,SUPPORT>`, launched
// once per level by host::forward (:440-470) with SUPPORT = level+1; level 0 overwrites the output
// with alpha = 1 (:47-60), later levels read-modify-write rgb (:62-106) and therefore depend on
// same-stream ordering (:224-226).
//
// Here all L levels run in ONE launch: a workgroup stages the noisy tile (halo = L) and the L
// guidance tiles in LDS once, every thread walks level 0..L-1 for its pixel and adds the level
// results in registers in level order -- the same fp32 sequence as L read-modify-write passes, so
// the output is bit-identical to the level-by-level statement (oracle/rto_oracle.c orc_filter) while
// the noisy image is read once instead of L times and the output is written once.
//
// Per level, per pixel p (support S = level+1, window (2S+1)^2, row-major tap order):
//   m = max_q g(q);  k_q = exp(g(q) - m);  out += (sum_q k_q rgb(q)) * (w(p) / sum_q k_q)
// Out-of-image taps: rgb = 0, g = -FLT_MAX (:140-143).
//
// Roofline: VALU-bound, not HBM-bound -- (2S+1)^2 taps of ~13 fp32 instructions each (the packed exp
// is 9 of them) against 48 B of traffic per pixel.  The tap loops are kept rolled per window row so
// the kernel stays at <= 128 VGPRs (4 waves/SIMD); fully unrolled it needs 256 VGPRs and runs one
// wave per SIMD.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "rto_launch.h"

#pragma clang fp contract(off)

namespace rto {

constexpr int kFiltW = 32, kFiltH = 8;  // output tile per 256-thread workgroup

// what Filtering::forward keeps for backward when requires_grad (filtering.cu:205-216)
struct LevelSave {
    float4 rgb_filtered;  // (sum_q k_q rgb_q) / sum_q k_q, alpha 0 (torch::zeros, never written)
    float max_val, inv_kernel_sum;
};

template <int S, int TW>
RTO_DEV void filter_level(const float* __restrict__ g, const float4* __restrict__ rgb, int centre, float w_pix,
                          float& o0, float& o1, float& o2, bool first, LevelSave* save = nullptr) {
    float max_val = -3.402823466e+38f;
#pragma unroll 1
    for (int dy = -S; dy <= S; ++dy) {
        const float* row = g + centre + dy * TW;
#pragma unroll
        for (int dx = -S; dx <= S; ++dx) max_val = fmaxf(max_val, row[dx]);
    }
    // accumulators as two packed pairs: {r, g} and {b, kernel_sum}.  The tile's alpha is staged as
    // 1, so kernel_sum += k is the lane fma(1, k, kernel_sum) = kernel_sum + k exactly, and each tap
    // costs two v_pk_fma_f32 instead of three FMAs and an add.
    float2v rg = {0.f, 0.f}, bs = {0.f, 0.f};
#pragma unroll 1
    for (int dy = -S; dy <= S; ++dy) {
        const int e0 = centre + dy * TW;
        // the exps of two neighbouring taps share packed fp32 instructions; the sums stay in tap order
#pragma unroll
        for (int dx = -S; dx < S; dx += 2) {
            float2v x2;
            x2.x = g[e0 + dx] - max_val;
            x2.y = g[e0 + dx + 1] - max_val;
            const float2v k2 = fexp_f32_le88_x2(x2);
            const float4 t0 = rgb[e0 + dx], t1 = rgb[e0 + dx + 1];
            // (explicit FMA, as in the oracle: nvcc contracts the reference's `rgba.x += t_rgb.x * k`
            // the same way)
            rg = __builtin_elementwise_fma(float2v{t0.x, t0.y}, float2v{k2.x, k2.x}, rg);
            bs = __builtin_elementwise_fma(float2v{t0.z, t0.w}, float2v{k2.x, k2.x}, bs);
            rg = __builtin_elementwise_fma(float2v{t1.x, t1.y}, float2v{k2.y, k2.y}, rg);
            bs = __builtin_elementwise_fma(float2v{t1.z, t1.w}, float2v{k2.y, k2.y}, bs);
        }
        {  // the window is 2S+1 wide: one tap left
            const float k = fexp_f32_le88(g[e0 + S] - max_val);
            const float4 t = rgb[e0 + S];
            rg = __builtin_elementwise_fma(float2v{t.x, t.y}, float2v{k, k}, rg);
            bs = __builtin_elementwise_fma(float2v{t.z, t.w}, float2v{k, k}, bs);
        }
    }
    float r = rg.x, gg = rg.y, b = bs.x;
    const float kernel_sum = bs.y;
    const float inv = 1.0f / kernel_sum;
    if (save) {
        save->max_val = max_val;
        save->inv_kernel_sum = inv;
        save->rgb_filtered = make_float4(r * inv, gg * inv, b * inv, 0.f);
    }
    const float w = w_pix * inv;
    r *= w;
    gg *= w;
    b *= w;
    if (first) {
        o0 = r;
        o1 = gg;
        o2 = b;
    } else {
        o0 += r;
        o1 += gg;
        o2 += b;
    }
}

// SAVE: the training forward (Filtering::forward with requires_grad, filtering.cu:596-665) -- the same
// pass, also storing rgb_filtered [n][L][H][W] (float4), max_map and inv_kernel_sum [n][L][H][W]
template <int L, bool SAVE>
__global__ void __launch_bounds__(256, 4) filter_fused(const float* __restrict__ weight,    // [n][L][H][W]
                                                        const float* __restrict__ guidance,  // [n][L][H][W]
                                                        const float4* __restrict__ img_in,   // [n][H][W]
                                                        float4* __restrict__ img_out,        // [n][H][W]
                                                        int H, int W, float4* __restrict__ rgb_filtered,
                                                        float* __restrict__ max_map, float* __restrict__ inv_kernel_sum) {
    constexpr int TW = kFiltW + 2 * L, TH = kFiltH + 2 * L;
    __shared__ float4 s_rgb[TH * TW];
    __shared__ float s_g[L][TH * TW];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * kFiltW - L, y0 = blockIdx.y * kFiltH - L;
    const int64_t HW = (int64_t)H * W;
    // image blockIdx.z of the batch
    weight += (int64_t)blockIdx.z * L * HW;
    guidance += (int64_t)blockIdx.z * L * HW;
    img_in += (int64_t)blockIdx.z * HW;
    img_out += (int64_t)blockIdx.z * HW;

    for (int e = tid; e < TH * TW; e += 256) {
        const int ty = e / TW, tx = e - ty * TW;
        const int gx = x0 + tx, gy = y0 + ty;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        const int64_t gi = (int64_t)gy * W + gx;
        float4 t = in ? img_in[gi] : make_float4(0.f, 0.f, 0.f, 0.f);
        t.w = 1.f;  // not an image value here: the multiplier of k in the kernel_sum lane (filter_level)
        s_rgb[e] = t;
#pragma unroll
        for (int l = 0; l < L; ++l) s_g[l][e] = in ? guidance[l * HW + gi] : -3.402823466e+38f;
    }
    __syncthreads();

    const int lx = tid & (kFiltW - 1), ly = tid / kFiltW;
    const int px = blockIdx.x * kFiltW + lx, py = blockIdx.y * kFiltH + ly;
    if (px >= W || py >= H) return;
    const int64_t pidx = (int64_t)py * W + px;
    const int centre = (ly + L) * TW + lx + L;

    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    LevelSave sv;
    auto level = [&](auto s_tag, int l, bool first) {
        constexpr int S = decltype(s_tag)::value;
        filter_level<S, TW>(s_g[l], s_rgb, centre, weight[l * HW + pidx], o0, o1, o2, first, SAVE ? &sv : nullptr);
        if constexpr (SAVE) {
            const int64_t si = ((int64_t)blockIdx.z * L + l) * HW + pidx;
            rgb_filtered[si] = sv.rgb_filtered;
            max_map[si] = sv.max_val;
            inv_kernel_sum[si] = sv.inv_kernel_sum;
        }
    };
    level(std::integral_constant<int, 1>{}, 0, true);
    if constexpr (L >= 2) level(std::integral_constant<int, 2>{}, 1, false);
    if constexpr (L >= 3) level(std::integral_constant<int, 3>{}, 2, false);
    if constexpr (L >= 4) level(std::integral_constant<int, 4>{}, 3, false);
    if constexpr (L >= 5) level(std::integral_constant<int, 5>{}, 4, false);
    if constexpr (L >= 6) level(std::integral_constant<int, 6>{}, 5, false);
    img_out[pidx] = make_float4(o0, o1, o2, 1.0f);
}

// Backward of the filter (filtering.cu:230-301, 667-707), gather form: one thread per pixel q and level,
//   grad_weight[l][q]   = <grad_out[q], rgb_filtered_l[q]>
//   grad_guidance[l][q] = sum_{p in window_S(q), p in image} w_l[p] * (exp(g_l[q] - max_l[p]) * inv_l[p])
//                                                             * <grad_out[p], img_in[q] - rgb_filtered_l[p]>
// with p in row-major order -- the order oracle/rto_oracle.c orc_filter_backward fixes (the reference
// scatters the same terms with atomicAdd in hardware order).  Tile + halo of grad_out and, per level, of
// rgb_filtered and {w, max, inv} are staged in LDS; q lies in every window it is gathered from, so
// g[q] - max[p] <= 0 and the branch-free exp applies.
template <int L>
__global__ void __launch_bounds__(256, 2) filter_backward(const float4* __restrict__ grad_out,      // [n][H][W]
                                                           const float4* __restrict__ img_in,        // [n][H][W]
                                                           const float* __restrict__ weight,         // [n][L][H][W]
                                                           const float* __restrict__ guidance,       // [n][L][H][W]
                                                           const float4* __restrict__ rgb_filtered,  // [n][L][H][W]
                                                           const float* __restrict__ max_map,        // [n][L][H][W]
                                                           const float* __restrict__ inv_kernel_sum, // [n][L][H][W]
                                                           float* __restrict__ grad_weight,          // [n][L][H][W]
                                                           float* __restrict__ grad_guidance,        // [n][L][H][W]
                                                           int H, int W) {
    constexpr int TW = kFiltW + 2 * L, TH = kFiltH + 2 * L;
    __shared__ float4 s_go[TH * TW];   // grad_out
    __shared__ float4 s_f[TH * TW];    // rgb_filtered of the level (w = 1 inside the image, 0 outside)
    __shared__ float4 s_wmi[TH * TW];  // {weight, max, inv_kernel_sum, -}

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * kFiltW - L, y0 = blockIdx.y * kFiltH - L;
    const int64_t HW = (int64_t)H * W;
    const int64_t img0 = (int64_t)blockIdx.z * HW, lvl0 = (int64_t)blockIdx.z * L * HW;

    const int lx = tid & (kFiltW - 1), ly = tid / kFiltW;
    const int qx = blockIdx.x * kFiltW + lx, qy = blockIdx.y * kFiltH + ly;
    const bool q_in = qx < W && qy < H;
    const int64_t qidx = (int64_t)qy * W + qx;
    const int centre = (ly + L) * TW + lx + L;
    const float4 in_q = q_in ? img_in[img0 + qidx] : make_float4(0.f, 0.f, 0.f, 0.f);

    for (int e = tid; e < TH * TW; e += 256) {
        const int ty = e / TW, tx = e - ty * TW;
        const int gx = x0 + tx, gy = y0 + ty;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        s_go[e] = in ? grad_out[img0 + (int64_t)gy * W + gx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        const int S = l + 1;
        __syncthreads();  // previous level's taps are done with s_f / s_wmi (and s_go is complete)
        for (int e = tid; e < TH * TW; e += 256) {
            const int ty = e / TW, tx = e - ty * TW;
            const int gx = x0 + tx, gy = y0 + ty;
            const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
            const int64_t gi = lvl0 + l * HW + (int64_t)gy * W + gx;
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f), wmi = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in) {
                f = rgb_filtered[gi];
                f.w = 1.f;  // marks "p is an image pixel" (the reference has no thread for the others)
                wmi = make_float4(weight[gi], max_map[gi], inv_kernel_sum[gi], 0.f);
            }
            s_f[e] = f;
            s_wmi[e] = wmi;
        }
        __syncthreads();
        if (q_in) {
            const int64_t qi = lvl0 + l * HW + qidx;
            {  // grad_weight_accumulate :244-247
                const float4 go = s_go[centre], f = s_f[centre];
                float t = go.x * f.x;
                t = __builtin_fmaf(go.y, f.y, t);
                t = __builtin_fmaf(go.z, f.z, t);
                grad_weight[qi] = t;
            }
            const float gq = guidance[qi];
            float acc = 0.f;
#pragma unroll 1
            for (int dy = -S; dy <= S; ++dy) {
                const int e0 = centre + dy * TW;
#pragma unroll 1
                for (int dx = -S; dx <= S; ++dx) {
                    const float4 f = s_f[e0 + dx];
                    const float4 wmi = s_wmi[e0 + dx];
                    const float4 go = s_go[e0 + dx];
                    const float k = fexp_f32_le88(gq - wmi.y) * wmi.z;  // :293
                    float res = go.x * (in_q.x - f.x);                // :294-297
                    res = __builtin_fmaf(go.y, in_q.y - f.y, res);
                    res = __builtin_fmaf(go.z, in_q.z - f.z, res);
                    res *= wmi.x * k;                                  // :298
                    if (f.w != 0.f) acc += res;                        // :300 (gathered, row-major p)
                }
            }
            grad_guidance[qi] = acc;
        }
    }
}

hipError_t launch_filter(const float* weight, const float* guidance, int L, int H, int W, int n, const float* img_in,
                         float* img_out, hipStream_t stream) {
    return launch_filter_train(weight, guidance, L, H, W, n, img_in, img_out, nullptr, nullptr, nullptr, stream);
}

// rgb_filtered == nullptr: inference forward; else all three save arrays are written too
hipError_t launch_filter_train(const float* weight, const float* guidance, int L, int H, int W, int n,
                               const float* img_in, float* img_out, float* rgb_filtered, float* max_map,
                               float* inv_kernel_sum, hipStream_t stream) {
    const dim3 grid((W + kFiltW - 1) / kFiltW, (H + kFiltH - 1) / kFiltH, n), block(256);
    const float4* in4 = reinterpret_cast<const float4*>(img_in);
    float4* out4 = reinterpret_cast<float4*>(img_out);
    float4* rf4 = reinterpret_cast<float4*>(rgb_filtered);
#define RTO_FILT(LL)                                                                                                   \
    case LL:                                                                                                           \
        if (rgb_filtered)                                                                                              \
            hipLaunchKernelGGL((filter_fused<LL, true>), grid, block, 0, stream, weight, guidance, in4, out4, H, W, rf4, \
                               max_map, inv_kernel_sum);                                                               \
        else                                                                                                           \
            hipLaunchKernelGGL((filter_fused<LL, false>), grid, block, 0, stream, weight, guidance, in4, out4, H, W,    \
                               (float4*)nullptr, (float*)nullptr, (float*)nullptr);                                    \
        break;
    switch (L) {  // kernel_apply filtering.cu:338-367 supports SUPPORT 1..6
        RTO_FILT(1) RTO_FILT(2) RTO_FILT(3) RTO_FILT(4) RTO_FILT(5) RTO_FILT(6)
        default: return hipErrorInvalidValue;
    }
#undef RTO_FILT
    return hipGetLastError();
}

hipError_t launch_filter_backward(const float* grad_out, const float* img_in, const float* weight, const float* guidance,
                                  const float* rgb_filtered, const float* max_map, const float* inv_kernel_sum, int L,
                                  int H, int W, int n, float* grad_weight, float* grad_guidance, hipStream_t stream) {
    const dim3 grid((W + kFiltW - 1) / kFiltW, (H + kFiltH - 1) / kFiltH, n), block(256);
    const float4* go4 = reinterpret_cast<const float4*>(grad_out);
    const float4* in4 = reinterpret_cast<const float4*>(img_in);
    const float4* rf4 = reinterpret_cast<const float4*>(rgb_filtered);
#define RTO_FILTB(LL)                                                                                                  \
    case LL:                                                                                                           \
        hipLaunchKernelGGL(filter_backward<LL>, grid, block, 0, stream, go4, in4, weight, guidance, rf4, max_map,      \
                           inv_kernel_sum, grad_weight, grad_guidance, H, W);                                          \
        break;
    switch (L) {
        RTO_FILTB(1) RTO_FILTB(2) RTO_FILTB(3) RTO_FILTB(4) RTO_FILTB(5) RTO_FILTB(6)
        default: return hipErrorInvalidValue;
    }
#undef RTO_FILTB
    return hipGetLastError();
}

}  // namespace rto
