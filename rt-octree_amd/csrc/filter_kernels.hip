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

#include "rto_launch.h"

#pragma clang fp contract(off)

namespace rto {

constexpr int kFiltW = 32, kFiltH = 8;  // output tile per 256-thread workgroup

template <int S, int TW>
RTO_DEV void filter_level(const float* __restrict__ g, const float4* __restrict__ rgb, int centre, float w_pix,
                          float& o0, float& o1, float& o2, bool first) {
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

template <int L>
__global__ void __launch_bounds__(256, 4) filter_fused(const float* __restrict__ weight,    // [n][L][H][W]
                                                        const float* __restrict__ guidance,  // [n][L][H][W]
                                                        const float4* __restrict__ img_in,   // [n][H][W]
                                                        float4* __restrict__ img_out,        // [n][H][W]
                                                        int H, int W) {
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
    filter_level<1, TW>(s_g[0], s_rgb, centre, weight[pidx], o0, o1, o2, true);
    if constexpr (L >= 2) filter_level<2, TW>(s_g[1], s_rgb, centre, weight[HW + pidx], o0, o1, o2, false);
    if constexpr (L >= 3) filter_level<3, TW>(s_g[2], s_rgb, centre, weight[2 * HW + pidx], o0, o1, o2, false);
    if constexpr (L >= 4) filter_level<4, TW>(s_g[3], s_rgb, centre, weight[3 * HW + pidx], o0, o1, o2, false);
    if constexpr (L >= 5) filter_level<5, TW>(s_g[4], s_rgb, centre, weight[4 * HW + pidx], o0, o1, o2, false);
    if constexpr (L >= 6) filter_level<6, TW>(s_g[5], s_rgb, centre, weight[5 * HW + pidx], o0, o1, o2, false);
    img_out[pidx] = make_float4(o0, o1, o2, 1.0f);
}

hipError_t launch_filter(const float* weight, const float* guidance, int L, int H, int W, int n, const float* img_in,
                         float* img_out, hipStream_t stream) {
    const dim3 grid((W + kFiltW - 1) / kFiltW, (H + kFiltH - 1) / kFiltH, n), block(256);
    const float4* in4 = reinterpret_cast<const float4*>(img_in);
    float4* out4 = reinterpret_cast<float4*>(img_out);
    switch (L) {  // kernel_apply filtering.cu:338-367 supports SUPPORT 1..6
        case 1: hipLaunchKernelGGL(filter_fused<1>, grid, block, 0, stream, weight, guidance, in4, out4, H, W); break;
        case 2: hipLaunchKernelGGL(filter_fused<2>, grid, block, 0, stream, weight, guidance, in4, out4, H, W); break;
        case 3: hipLaunchKernelGGL(filter_fused<3>, grid, block, 0, stream, weight, guidance, in4, out4, H, W); break;
        case 4: hipLaunchKernelGGL(filter_fused<4>, grid, block, 0, stream, weight, guidance, in4, out4, H, W); break;
        case 5: hipLaunchKernelGGL(filter_fused<5>, grid, block, 0, stream, weight, guidance, in4, out4, H, W); break;
        case 6: hipLaunchKernelGGL(filter_fused<6>, grid, block, 0, stream, weight, guidance, in4, out4, H, W); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace rto
