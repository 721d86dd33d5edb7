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
#include "rto_denoise_launch.h"

#pragma clang fp contract(off)

namespace rto {

#ifndef RTO_FILTER_PK
#define RTO_FILTER_PK 0
#endif
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
    // accumulators {r, g}, {b, kernel_sum}.  The tile's alpha is staged as 1, so kernel_sum += k is fma(1, k, kernel_sum) =
    // kernel_sum + k exactly: four FMAs per tap.
    // Scalar FMAs (v_fma_f32), one tap at a time.  Rounds 1-2 ran this loop on packed pairs -- v_pk_fma_f32, exps of two
    // neighbouring taps at once (-DRTO_FILTER_PK=1 rebuilds it) -- the same bits and the same speed (a packed instruction
    // issues at half rate), until round 3 found the packed build's bits wrong in lanes 48..63 of some waves whenever another
    // process kept the GPU busy with MFMA-dense kernels (DESIGN_HISTORY.md "Determinism when the GPU is shared").  Of all kernels only this one
    // held v_pk_fma_f32; v_pk_add / v_pk_mul (filter_fast, shading) never showed it.
#if !RTO_FILTER_PK
    float2v rg = {0.f, 0.f}, bs = {0.f, 0.f};
#pragma unroll 1
    for (int dy = -S; dy <= S; ++dy) {
        const int e0 = centre + dy * TW;
#pragma unroll
        for (int dx = -S; dx <= S; ++dx) {
            const float k = fexp_f32_le88(g[e0 + dx] - max_val);
            const float4 t = rgb[e0 + dx];
            rg.x = __builtin_fmaf(t.x, k, rg.x);
            rg.y = __builtin_fmaf(t.y, k, rg.y);
            bs.x = __builtin_fmaf(t.z, k, bs.x);
            bs.y = __builtin_fmaf(t.w, k, bs.y);
        }
    }
#elif RTO_FILTER_PK == 3  // (experiment: scalar exps, packed accumulation -- which half of the packed form is it?)
    float2v rg = {0.f, 0.f}, bs = {0.f, 0.f};
#pragma unroll 1
    for (int dy = -S; dy <= S; ++dy) {
        const int e0 = centre + dy * TW;
#pragma unroll
        for (int dx = -S; dx <= S; ++dx) {
            const float k = fexp_f32_le88(g[e0 + dx] - max_val);
            const float4 t = rgb[e0 + dx];
            rg = __builtin_elementwise_fma(float2v{t.x, t.y}, float2v{k, k}, rg);
            bs = __builtin_elementwise_fma(float2v{t.z, t.w}, float2v{k, k}, bs);
        }
    }
#elif RTO_FILTER_PK == 2  // (experiment: packed exps, scalar accumulation)
    float2v rg = {0.f, 0.f}, bs = {0.f, 0.f};
#pragma unroll 1
    for (int dy = -S; dy <= S; ++dy) {
        const int e0 = centre + dy * TW;
#pragma unroll
        for (int dx = -S; dx <= S; dx += 2) {
            float2v x2;
            x2.x = g[e0 + dx] - max_val;
            x2.y = dx + 1 <= S ? g[e0 + dx + 1] - max_val : 0.f;
            const float2v k2 = fexp_f32_le88_x2(x2);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (dx + h > S) break;
                const float k = h ? k2.y : k2.x;
                const float4 t = rgb[e0 + dx + h];
                rg.x = __builtin_fmaf(t.x, k, rg.x);
                rg.y = __builtin_fmaf(t.y, k, rg.y);
                bs.x = __builtin_fmaf(t.z, k, bs.x);
                bs.y = __builtin_fmaf(t.w, k, bs.y);
            }
        }
    }
#else
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
#endif
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

// Empty-space culling carried into the filter (round 3): the render context's tile marks (FrameBatch::tile_mask: bit t = the
// 8x8 render tile t may hold a ray that meets density; last word = keep the whole frame) and the filter's output tile for a
// workgroup whose whole neighbourhood is background.  mask == nullptr: off.
struct FilterCull {
    const uint32_t* mask;
    int mask_words, tiles_x;
    const float4* fill;  // [tile height][tile width]: the output tile of a workgroup that sees only background
    // SPARSE lean frames (round 6; filter_fast on packed maps only): img_in holds no pixel of an unmarked render tile -- it is the
    // background (bg, bg, bg) -- and the packed maps none of a network tile that saw only background (guidance_fused skipped
    // it): its 8 fp16 values are fill_maps.  The staging loop substitutes both.
    int sparse;
    float bg;
    uint4 fill_maps;
};

constexpr int kMapHalo = 2;  // a GuidanceNet map value depends on the 5x5 aux pixels around it (two 3x3 convolutions)

// A workgroup whose staged region (outputs + halo, RW x RH pixels from (rx0, ry0)), grown by the network's receptive field,
// lies inside the image and inside culled render tiles reads nothing but background pixels and the network's background
// maps.  Its arithmetic is then the same as that of any other such workgroup, thread for thread: the output tile is the one
// the same kernel produced once on a synthetic background image (FilterCull::fill).  Workgroup-uniform; every thread calls.
template <int SW, int SH>
__device__ __forceinline__ bool sees_only_background(const FilterCull& cull, int sx0, int sy0, int H, int W) {
    constexpr int RW = SW + 2 * kMapHalo, RH = SH + 2 * kMapHalo;
    const int rx0 = sx0 - kMapHalo, ry0 = sy0 - kMapHalo;
    if (!(rx0 >= 0 && ry0 >= 0 && rx0 + RW <= W && ry0 + RH <= H)) return false;  // (the zero padding is not background)
    const uint32_t* fm = cull.mask + (size_t)blockIdx.z * cull.mask_words;
    const int tx0 = rx0 >> 3, ty0 = ry0 >> 3, nx = ((rx0 + RW - 1) >> 3) - tx0 + 1, ny = ((ry0 + RH - 1) >> 3) - ty0 + 1;
    int any = 0;
    if ((int)threadIdx.x < nx * ny) {
        const int ty = ty0 + (int)threadIdx.x / nx, tx = tx0 + (int)threadIdx.x % nx;
        const uint32_t t = (uint32_t)(ty * cull.tiles_x + tx);
        any = (int)(((fm[t >> 5] >> (t & 31u)) | fm[cull.mask_words - 1]) & 1u);
    }
    return !__syncthreads_or(any);
}

// SAVE: the training forward (Filtering::forward with requires_grad, filtering.cu:596-665) -- the same
// pass, also storing rgb_filtered [n][L][H][W] (float4), max_map and inv_kernel_sum [n][L][H][W]
template <int L, bool SAVE>
__global__ void __launch_bounds__(256, 4) filter_fused(const float* __restrict__ weight,    // [n][L][H][W]
                                                        const float* __restrict__ guidance,  // [n][L][H][W]
                                                        const float4* __restrict__ img_in,   // [n][H][W]
                                                        float4* __restrict__ img_out,        // [n][H][W]
                                                        int H, int W, float4* __restrict__ rgb_filtered,
                                                        float* __restrict__ max_map, float* __restrict__ inv_kernel_sum,
                                                        const FilterCull cull) {
    constexpr int TW = kFiltW + 2 * L, TH = kFiltH + 2 * L;
    if (!SAVE && cull.mask) {
        if (sees_only_background<TW, TH>(cull, (int)blockIdx.x * kFiltW - L, (int)blockIdx.y * kFiltH - L, H, W)) {
            const int lx = threadIdx.x & (kFiltW - 1), ly = threadIdx.x / kFiltW;
            img_out[((int64_t)blockIdx.z * H + blockIdx.y * kFiltH + ly) * W + blockIdx.x * kFiltW + lx] = cull.fill[ly * kFiltW + lx];
            return;
        }
    }
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


// ------------------------------------------------------------------ factorised forward (tolerance path)
// filter_fast<L>: the same filter with the exponentials factorised.  For a pixel p and level l
//     out_l(p) = w_l(p) * sum_q e^{g(q) - m_p} rgb(q) / sum_q e^{g(q) - m_p}
// and the factor e^{-m_p} cancels: any constant c common to the window gives the same quotient.  With
// c_l = the maximum of g_l over the workgroup's staged tile, E_l(q) = e^{g_l(q) - c_l} is computed ONCE per
// staged pixel (not once per tap: 4 exps per pixel instead of 164 at L = 4) and the level becomes a box
// filter of P_l = {E r, E g, E b, E}:  out_l(p) = w_l(p) * box(P_l).rgb / box(P_l).w.  Each thread owns 4
// vertically adjacent pixels, so a window row's sum (2S+1 LDS reads) is shared by up to 4 outputs:
// 59 16-byte LDS reads per pixel instead of 164 + 164, and two packed adds per read instead of an exp and
// two packed FMAs.  Not bit-identical to filter_fused (products are rounded before they are summed, the sums
// run in another order, v_exp_f32 replaces the polynomial): relative differences of ~1e-6, > 120 dB
// between the two routes -- the north-star's tolerance for float paths is 1e-4 dB.  filter_fused stays the
// bit-exact route (tests, CLI default) and the training forward.
// Guard: E must not underflow inside a window.  GuidanceNet ends in ReLU6, so g lies in [0, 6]; for
// arbitrary maps a tile whose in-image range of g_l exceeds 80 takes the per-pixel-maximum route below
// (workgroup-uniform branch, taps read from global memory: slow, correct).
// Outputs per workgroup: 32 x RTO_FAST_H.  Rounds 2-3: 32 x 32 (51 KB of LDS, 3 workgroups per CU).  Round 4: 32 x 16 (31 KB, 4 per CU) --
// since the culling only a third of the tiles is computed and those pay one exposed memory latency each, so residency counts for more
// than the halo re-reads (1.56 -> 1.88 staged pixels per output), and a finer tile also qualifies as "sees only background" more often:
// 0.937 -> 0.856 ms per 100 frames of the bench scene in one box (profiles/r4_y_ab_filter_tile.txt).  The fill tile (rto_guidance_abi.cpp
// ensure_fill_tile: 32 x 32 measured on a synthetic frame) covers two such workgroups; a workgroup copies its first RTO_FAST_H rows.
#ifndef RTO_FAST_H
#define RTO_FAST_H 16
#endif
#ifndef RTO_FAST_WGS
#define RTO_FAST_WGS 4
#endif
constexpr int kFastW = 32, kFastH = RTO_FAST_H, kFastRows = RTO_FAST_H / 8;  // outputs per workgroup; rows per thread
constexpr int kFastRowStride = 48, kFastHalf = 24;  // LDS row of a staged tile: [even columns | pad | odd columns | pad] (filter_fast)

// a level of filter_fast for a tile whose guidance range would underflow the factorised exponentials:
// per-pixel maximum as in the exact form, taps from global memory (rare, slow, correct)
// (half_stride != 0: g points at fp16 values half_stride halves apart -- the packed maps)
__device__ __noinline__ float4 filter_level_wide(const float* __restrict__ g_, const float4* __restrict__ img_in, int S, int H,
                                                 int W, int px, int py, float w, int half_stride) {
    if (px >= W || py >= H) return make_float4(0.f, 0.f, 0.f, 0.f);
    auto g = [&](int64_t i) -> float {
        return half_stride ? (float)reinterpret_cast<const _Float16*>(g_)[i * half_stride] : g_[i];
    };
    float m = -3.402823466e+38f;
    for (int dy = -S; dy <= S; ++dy)
        for (int dx = -S; dx <= S; ++dx) {
            const int qx = px + dx, qy = py + dy;
            if (qx >= 0 && qx < W && qy >= 0 && qy < H) m = fmaxf(m, g((int64_t)qy * W + qx));
        }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ks = 0.f;
    for (int dy = -S; dy <= S; ++dy)
        for (int dx = -S; dx <= S; ++dx) {
            const int qx = px + dx, qy = py + dy;
            if (qx >= 0 && qx < W && qy >= 0 && qy < H) {
                const int64_t qi = (int64_t)qy * W + qx;
                const float k = __builtin_amdgcn_exp2f((g(qi) - m) * 1.44269504088896340736f);
                const float4 t = img_in[qi];
                a0 += k * t.x;
                a1 += k * t.y;
                a2 += k * t.z;
                ks += k;
            }
        }
    const float ww = w / ks;
    return make_float4(a0 * ww, a1 * ww, a2 * ww, 0.f);
}

// PACKED (L = 4): the maps arrive as the GuidanceNet kernel's packed output, fp16 [n][H][W][8] = 4 softmax logits + 4
// guidance values per pixel (`weight` points at it, `guidance` is unused): one 16-byte load per staged pixel
// instead of 4 strided dword loads, the weights by softmax_weights4 on the logits -- the same values as the fp32
// maps hold, so the same output bit for bit.
// Round 5 (VERDICT r4 task 6): a workgroup walks a STRIP of kFastStrip tiles along x with the next computed tile's pixels and
// maps on their way into registers while the current tile's levels run -- since the culling a third of the tiles is computed and
// each paid one exposed memory latency with 4 waves per SIMD to hide it (62 % of the wave-cycles waiting).  The noisy pixels
// never needed LDS: staged element e is written and read by the same thread, so they stay in registers and the workgroup's LDS
// is the P_l tile alone (15 instead of 31 KB).
#ifndef RTO_FAST_STRIP
#define RTO_FAST_STRIP 5
#endif
constexpr int kFastStrip = RTO_FAST_STRIP;
// (the fp32-plane form with 5 or 6 levels holds up to 36 map values per thread: built for 3 workgroups per CU, no spills)
template <int L, bool PACKED>
__global__ void __launch_bounds__(256, (PACKED || L <= 4) ? RTO_FAST_WGS : 3) filter_fast(const float* __restrict__ weight,    // [n][L][H][W]
                                                     const float* __restrict__ guidance,  // [n][L][H][W]
                                                     const float4* __restrict__ img_in,   // [n][H][W]
                                                     float4* __restrict__ img_out,        // [n][H][W]
                                                     int H, int W, const FilterCull cull) {
    constexpr int SW = kFastW + 2 * L, SH = kFastH + 2 * L, NE = SW * SH;
    constexpr int PER = (NE + 255) / 256;  // staged elements per thread
    // LDS rows of P_l and of the window-row sums are SPLIT BY COLUMN PARITY: column c of a row sits at (c & 1) * kFastHalf +
    // (c >> 1) of a kFastRowStride-slot row.  Pass A of the box filter gives a thread TWO adjacent output columns (it reads
    // 2 S + 2 values for them instead of 2 (2 S + 1)): its lanes then walk one parity half with stride 1 -- conflict-free
    // 16-byte reads (a half offset of 24 slots = 128 B mod 256 B keeps the alternating lanes of pass B and of the stores on
    // different banks too).
    constexpr int SWP = kFastRowStride, HALF = kFastHalf;
    static_assert(SW / 2 <= HALF && HALF + SW / 2 <= SWP, "a parity half holds the even / odd columns of a staged row");
    extern __shared__ float4 s_dyn[];
    float4* s_p = s_dyn;               // [SH][SWP] P_l of the current level
    float4* s_hs = s_dyn + SH * SWP;   // [kFastH + 2 L][SWP] window-row sums of the current level (two-pass box filter)
    float4* s_rgb = s_hs + (kFastH + 2 * L) * SWP;  // fp32-plane form only: [NE] noisy tile (the packed form keeps its pixels in registers)
    __shared__ float s_red[2][4];

    const int tid = threadIdx.x;
    const int tiles_x = (W + kFastW - 1) / kFastW;
    // the strip's length follows from the launch: ceil(tiles_x / gridDim.x) <= kFastStrip.  A batch of frames is launched with
    // full strips (the prefetch pays); a LONE frame with as short ones as it takes to put a few workgroups on every CU -- 250
    // workgroups of 5 tiles each left three quarters of the chip idle while each walked its strip (fast_strip_for)
    const int strip = (tiles_x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tx_first = blockIdx.x * strip;
    const int y0 = blockIdx.y * kFastH - L;
    const int64_t HW = (int64_t)H * W;
    weight += (int64_t)blockIdx.z * L * HW;  // (PACKED: 8 halves = 4 floats per pixel = L * HW floats per image as well)
    guidance += (int64_t)blockIdx.z * L * HW;
    img_in += (int64_t)blockIdx.z * HW;
    img_out += (int64_t)blockIdx.z * HW;
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    const _Float16* packed = reinterpret_cast<const _Float16*>(weight);
    const int lx = tid & (kFastW - 1), ry = tid / kFastW;  // column, row group
    const int py0 = blockIdx.y * kFastH + ry * kFastRows;

    // Everything the culling needs -- which tiles of the strip see only background (filled from the measured tile, not computed:
    // bit ts of `skip`), and for sparse frames which network tiles under the staged region stored no maps (`nskip`) and which
    // render tiles stored their pixels (`rm`) -- follows from ONE bitmap of render-tile marks: 6 tile rows from R0 = (y0 >> 3) - 1,
    // 32 columns from C0 = 4 tx_first - 5, bit c of rm[r].  Every wave builds it for itself (round 6): lane = one render tile,
    // three independent mark words per lane requested together, the rows by ballot -- no LDS, no barrier, one memory round trip.
    // (Before: a dependent load + barrier per strip tile, then up to ten dependent mark loads per network tile -- 8-25 k clocks of
    //  a computing workgroup's ~90 k went by before its first pixel load, profiles/r6_p_filter_stamps.txt.)
    uint32_t skip = 0, nskip = 0, rm[6] = {0, 0, 0, 0, 0, 0};
    const int R0 = (y0 >> 3) - 1, C0 = 4 * tx_first - 5;
    const int ntx0 = (tx_first * kFastW - L) >> 5, nty0 = y0 >> 3;  // first network tile (32 x 8) under the staged region
    static_assert(kFastW == 32 && kFastH == 16 && L <= 6 && kFastStrip <= 5, "6 rows x 32 columns of render tiles cover a strip's neighbourhoods");
    if (cull.mask) {
        const uint32_t* const fm = cull.mask + (size_t)blockIdx.z * cull.mask_words;
        const int ln = tid & 63;
        const uint32_t keep_all = fm[cull.mask_words - 1] & 1u;
        uint32_t word[3];
        bool inside[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // render tile (row R0 + 2 k + (lane >> 5), column C0 + (lane & 31))
            const int ty = R0 + 2 * k + (ln >> 5), tx = C0 + (ln & 31);
            inside[k] = tx >= 0 && ty >= 0 && tx < cull.tiles_x && ty * 8 < H;
            const uint32_t t = inside[k] ? (uint32_t)(ty * cull.tiles_x + tx) : 0u;
            word[k] = fm[t >> 5] >> (t & 31u);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned long long b = __builtin_amdgcn_ballot_w64(inside[k] && ((word[k] | keep_all) & 1u) != 0u);
            rm[2 * k] = (uint32_t)b;
            rm[2 * k + 1] = (uint32_t)(b >> 32);
        }
        // a tile sees only background when its staged region, grown by the network's receptive field, lies inside the image and in
        // unmarked render tiles (sees_only_background above, from the bitmap; all scalar)
#pragma unroll
        for (int ts = 0; ts < kFastStrip; ++ts) {
            constexpr int RW = SW + 2 * kMapHalo, RH = SH + 2 * kMapHalo;
            const int rx0 = (tx_first + ts) * kFastW - L - kMapHalo, ry0 = y0 - kMapHalo;
            if (ts < strip && tx_first + ts < tiles_x && rx0 >= 0 && ry0 >= 0 && rx0 + RW <= W && ry0 + RH <= H) {
                uint32_t m = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    if (R0 + r >= (ry0 >> 3) && R0 + r <= ((ry0 + RH - 1) >> 3)) m |= rm[r];
                const int lo = (rx0 >> 3) - C0, n = ((rx0 + RW - 1) >> 3) - (rx0 >> 3) + 1;  // 0 <= lo, lo + n <= 32
                if (((m >> lo) & ((1u << n) - 1u)) == 0u) skip |= 1u << ts;
            }
        }
        for (int ts = 0; ts < strip && tx_first + ts < tiles_x; ++ts)
            if ((skip >> ts) & 1u) {
#pragma unroll
                for (int r = 0; r < kFastRows; ++r)
                    img_out[(int64_t)(py0 + r) * W + (tx_first + ts) * kFastW + lx] = cull.fill[(ry * kFastRows + r) * kFastW + lx];
            }
        // sparse frames: guidance_fused's decision to skip a 32 x 8 output tile (its 36 x 12 input region inside the image and in
        // unmarked render tiles -- guidance_kernels.hip `skip_tiles`), restated for the network tile (ntx0 + (lane & 7), nty0 + (lane >> 3)) -- bit
        // lane of `nskip`; the staged region spans at most strip + 2 <= 7 columns and 4 rows of them
        if (PACKED && cull.sparse) {
            const int c = ln & 7, r = (ln >> 3) & 3, ntx = ntx0 + c, nty = nty0 + r;
            const int nx0 = ntx * 32 - 2, ny0 = nty * 8 - 2;
            // (its render-tile rows nty - 1 .. nty + 1 are rows r .. r + 2 of the bitmap, its columns 4 ntx - 1 .. 4 ntx + 4 bits 4 c ..)
            const uint32_t m0 = rm[0] | rm[1] | rm[2], m1 = rm[1] | rm[2] | rm[3], m2 = rm[2] | rm[3] | rm[4], m3 = rm[3] | rm[4] | rm[5];
            const uint32_t m = r == 0 ? m0 : r == 1 ? m1 : r == 2 ? m2 : m3;
            const bool sk = ln < 32 && c <= kFastStrip + 1 && ntx >= 0 && nty >= 0 && ntx * 32 < W && nty * 8 < H && nx0 >= 0 && ny0 >= 0 &&
                            nx0 + 36 <= W && ny0 + 12 <= H && ((m >> (4 * c)) & 0x3fu) == 0u;
            nskip = (uint32_t)__builtin_amdgcn_ballot_w64(sk);
        }
    }
    auto next_live = [&](int ts) {  // first tile >= ts of the strip that has to be computed (kFastStrip: none)
        while (ts < strip && tx_first + ts < tiles_x && ((skip >> ts) & 1u)) ++ts;
        return (ts < strip && tx_first + ts < tiles_x) ? ts : kFastStrip;
    };
    // what a tile's computation needs from memory, as it arrives: the staged noisy pixels, the guidance values of the staged
    // elements and the weights of this thread's outputs (packed: raw fp16 maps, converted when the tile's turn comes)
    // (packed: raw fp16 values -- of a staged element only its 4 guidance values, 8 bytes -- converted where they are used)
    struct Fetched {
        float4 rgb[PACKED ? PER : 1];
        half4_t hg[PACKED ? PER : 1];
        half4_t hw[PACKED ? kFastRows : 1];
        float gv[PACKED ? 1 : L][PACKED ? 1 : PER];
        float wl[PACKED ? 1 : L][PACKED ? 1 : kFastRows];
    };
    // staged element i of this thread, in STORAGE order (a row's even columns, then its odd ones: 8 consecutive lanes store to
    // 8 consecutive LDS slots): row ty, column tx of the staged tile, slot in s_p -- packed into ONE register per element
    // (slot | tx << 11 | ty << 17; bit 31: beyond the tile) and unpacked where it is used: the compiler otherwise keeps every
    // derived index of every element live across the strip loop, and the packed form spills
    uint32_t epk[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + i * 256;
        const int ty = e / SW, r = e - ty * SW, half = r >= SW / 2 ? 1 : 0, k = r - half * (SW / 2);
        epk[i] = (uint32_t)(ty * SWP + half * HALF + k) | (uint32_t)(2 * k + half) << 11 | (uint32_t)ty << 17 | (e < NE ? 0u : 0x80000000u);
    }
    static_assert(SH * SWP <= 2048 && SW <= 64 && SH <= 32, "bit budget of the packed element index");
    auto elem = [&](int i, int& ty, int& tx, bool& in_tile) {
        uint32_t v = epk[i];
        asm volatile("" : "+v"(v));  // (opaque: nothing derived from it is loop-invariant to the compiler)
        ty = (int)((v >> 17) & 31u);
        tx = (int)((v >> 11) & 63u);
        in_tile = (int32_t)v >= 0;
        return (int)(v & 2047u);
    };
    auto gindex_xy = [&](int x0, int i, int& gx, int& gy) {  // staged element i of this thread -> global index (or -1 outside the image)
        int ty, tx;
        bool in_tile;
        elem(i, ty, tx, in_tile);
        gx = x0 + tx;
        gy = y0 + ty;
        return (in_tile && gx >= 0 && gx < W && gy >= 0 && gy < H) ? gy * W + gx : -1;  // (a frame has < 2^31 pixels)
    };
    auto gindex = [&](int x0, int i) {
        int gx, gy;
        return gindex_xy(x0, i, gx, gy);
    };
    // Packed form, round 6: what decides a staged element's loads -- is it inside the image, was its pixel stored, were its maps
    // stored -- depends on the tile only through the tile's position in the strip, so it is worked out ONCE per strip into three
    // registers: bit 4 ts + i of `imgbits` = element i is inside the image when strip tile ts is staged, the same bit of `pixbits` =
    // its pixel is in memory, bit 8 i + ts of `mapbits` = its maps are.  A tile's fetch then tests bits.  (The first form re-derived
    // all of it per tile and element: ~280 of a tile's ~1080 vector instructions and a dozen branches per element.)
    uint32_t imgbits = 0, pixbits = 0, mapbits = 0;
    if constexpr (PACKED) {
        if (next_live(0) < kFastStrip) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                int ty, tx;
                bool in_tile;
                elem(i, ty, tx, in_tile);
                const int gy = y0 + ty, rr = (gy >> 3) - R0;  // (staged rows: bitmap rows 1..4)
                const bool yok = in_tile && gy >= 0 && gy < H;
                const uint32_t roww = rr == 1 ? rm[1] : rr == 2 ? rm[2] : rr == 3 ? rm[3] : rm[4];
                const int kc = ((tx - L) >> 3) + 5;  // bitmap column of this element in the strip's first tile; + 4 per tile
                const int kn = ((tx - L) >> 5) + 1;  // network-tile column of it there, from ntx0; + 1 per tile
                const uint32_t pb = cull.sparse ? (roww >> kc) & 0x11111u : 0x11111u;           // bits 4 ts
                const uint32_t nb = cull.sparse ? ~(nskip >> (8 * (rr - 1) + kn)) & 0x1fu : 0x1fu;  // bits ts
                uint32_t xs = 0, xb = 0;  // its column is inside the image: bits 4 ts / bits ts
#pragma unroll
                for (int ts = 0; ts < kFastStrip; ++ts) {
                    const int gx = (tx_first + ts) * kFastW - L + tx;
                    const uint32_t ok = yok && gx >= 0 && gx < W ? 1u : 0u;
                    xs |= ok << (4 * ts);
                    xb |= ok << ts;
                }
                imgbits |= xs << i;
                pixbits |= (pb & xs) << i;
                mapbits |= (nb & xb) << (8 * i);
            }
        }
    }
    static_assert(!PACKED || (PER <= 4 && kFastStrip <= 5), "bit budget of imgbits / pixbits / mapbits");
    // this thread's outputs sit in network-tile row (wave >> 1) + 1 of the region (kFastRows = 2 rows of one 8-row tile), column ts + 1
    const uint32_t outbits = (uint32_t)__builtin_amdgcn_readfirstlane((int)(~(nskip >> (8 * ((ry >> 2) + 1) + 1)) & 0x1fu));
    static_assert(kFastRows == 2 && kFastH == 16, "a thread's two output rows share a network tile; row = wave >> 1");
    auto fetch = [&](int tile, Fetched& f) {
        const int x0 = tile * kFastW - L, px = tile * kFastW + lx;
        if constexpr (PACKED) {
            // (sparse: whether a value exists in memory is known from registers BEFORE its load is issued, so the load is simply not
            //  issued for the lanes that take the constant -- a select on the loaded value would make this prefetch wait for it)
            const int ts = tile - tx_first;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                int ty, tx;
                bool in_tile;
                elem(i, ty, tx, in_tile);
                const int gi = (y0 + ty) * W + x0 + tx;  // (used only where the bits say it is inside; a frame has < 2^31 pixels)
                const bool in = ((imgbits >> (4 * ts + i)) & 1u) != 0u;
                const bool ld_rgb = ((pixbits >> (4 * ts + i)) & 1u) != 0u, ld_map = ((mapbits >> (8 * i + ts)) & 1u) != 0u;
                const float bgv = in ? cull.bg : 0.f;  // (only ever used when sparse)
                f.rgb[i] = make_float4(bgv, bgv, bgv, 0.f);
                if (ld_rgb) f.rgb[i] = img_in[gi];
                f.hg[i] = in ? __builtin_bit_cast(half4_t, make_uint2(cull.fill_maps.z, cull.fill_maps.w)) : half4_t{0, 0, 0, 0};
                if (ld_map) f.hg[i] = *reinterpret_cast<const half4_t*>(packed + (int64_t)gi * 8 + 4);
            }
            const bool out_stored = !cull.sparse || ((outbits >> ts) & 1u) != 0u;  // (wave-uniform)
#pragma unroll
            for (int r = 0; r < kFastRows; ++r) {
                const bool in = px < W && py0 + r < H;
                f.hw[r] = in ? __builtin_bit_cast(half4_t, make_uint2(cull.fill_maps.x, cull.fill_maps.y)) : half4_t{0, 0, 0, 0};
                if (in && out_stored) f.hw[r] = *reinterpret_cast<const half4_t*>(packed + ((int64_t)(py0 + r) * W + px) * 8);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int gi = gindex(x0, i);
                if (tid + i * 256 < NE) s_rgb[tid + i * 256] = gi >= 0 ? img_in[gi] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int l = 0; l < L; ++l) f.gv[l][i] = gi >= 0 ? guidance[l * HW + gi] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < kFastRows; ++r) {
                const bool in = px < W && py0 + r < H;
#pragma unroll
                for (int l = 0; l < L; ++l) f.wl[l][r] = in ? weight[l * HW + (int64_t)(py0 + r) * W + px] : 0.f;
            }
        }
    };
    // the fp32-plane form (up to 6 levels of maps per staged element) has no registers for a second tile in flight: it fetches
    // its tile when its turn comes, like before; the packed form -- the throughput route -- prefetches
    constexpr bool PIPE = PACKED;

    Fetched nxt;
    int ts_live = next_live(0);
    if (PIPE && ts_live < kFastStrip) fetch(tx_first + ts_live, nxt);
#pragma nounroll
    while (ts_live < kFastStrip) {  // (workgroup-uniform)
        const int tile = tx_first + ts_live;
        const int x0 = tile * kFastW - L, px = tile * kFastW + lx;
        ts_live = next_live(ts_live + 1);
        if (!PIPE) fetch(tile, nxt);
        // ---- this tile's values out of the fetch registers (the pixels' rgb only: the filter never reads their alpha)
        Fetched cur = nxt;
        float cr[PACKED ? PER : 1], cg[PACKED ? PER : 1], cb[PACKED ? PER : 1];
        if constexpr (PACKED) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                cr[i] = nxt.rgb[i].x;
                cg[i] = nxt.rgb[i].y;
                cb[i] = nxt.rgb[i].z;
            }
        }
        bool inimg[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) inimg[i] = PACKED ? ((imgbits >> (4 * (tile - tx_first) + i)) & 1u) != 0u : gindex(x0, i) >= 0;
        auto gval = [&](int l, int i) -> float {
            if constexpr (PACKED)
                return (float)cur.hg[i][l];
            else
                return cur.gv[l][i];
        };
        float wl_all[L][kFastRows];
        static_assert(!PACKED || L == 4, "packed maps hold 4 levels");
#pragma unroll
        for (int r = 0; r < kFastRows; ++r) {
            const bool in = px < W && py0 + r < H;
            if constexpr (PACKED) {
                float logit[4], wgt[4];
#pragma unroll
                for (int l = 0; l < 4; ++l) logit[l] = in ? (float)cur.hw[r][l] : 0.f;
                softmax_weights4(logit, wgt);
#pragma unroll
                for (int l = 0; l < L; ++l) wl_all[l][r] = in ? wgt[l] : 0.f;
            } else {
#pragma unroll
                for (int l = 0; l < L; ++l) wl_all[l][r] = cur.wl[PACKED ? 0 : l][PACKED ? 0 : r];
            }
        }
        // ---- the next computed tile's loads: in flight while this tile's levels run
        if (PIPE && ts_live < kFastStrip) fetch(tx_first + ts_live, nxt);

        float o[kFastRows][3];
#pragma unroll
        for (int r = 0; r < kFastRows; ++r) o[r][0] = o[r][1] = o[r][2] = 0.f;

        // ---- ONE constant for all levels of the tile (round 6): c = the maximum of every level's g over the image pixels of the
        // staged tile (any constant common to a window cancels; GuidanceNet's maps lie in [0, 6]), lo = the minimum -- one
        // reduction and one barrier per tile where each level had its own.  A tile whose values spread over more than 80 takes
        // the per-pixel-maximum route for all its levels.
        float mx = -3.402823466e+38f, mn = 3.402823466e+38f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (inimg[i]) {
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    mx = fmaxf(mx, gval(l, i));
                    mn = fminf(mn, gval(l, i));
                }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, d, 64));
            mn = fminf(mn, __shfl_xor(mn, d, 64));
        }
        if ((tid & 63) == 0) {
            s_red[0][tid >> 6] = mx;
            s_red[1][tid >> 6] = mn;
        }
        __syncthreads();  // (s_red is next written a tile -- several barriers -- later)
        const float c = fmaxf(fmaxf(s_red[0][0], s_red[0][1]), fmaxf(s_red[0][2], s_red[0][3]));
        const float lo = fminf(fminf(s_red[1][0], s_red[1][1]), fminf(s_red[1][2], s_red[1][3]));
        const bool wide = !(c - lo <= 80.f);  // workgroup-uniform (NaNs take the slow route too)

        auto level = [&](auto l_tag) {
            constexpr int l = decltype(l_tag)::value;
            // (s_p is free: every thread is past the barrier that ended the previous level's -- or tile's -- pass A, its last reader)
            if (!wide) {
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int e = tid + i * 256;
                    if (e < NE) {
                        const float E = inimg[i] ? __builtin_amdgcn_exp2f((gval(l, i) - c) * 1.44269504088896340736f) : 0.f;
                        int ty, tx;
                        bool in_tile;
                        const int sp = elem(i, ty, tx, in_tile);
                        if constexpr (PACKED) {
                            s_p[sp] = make_float4(E * cr[i], E * cg[i], E * cb[i], E);
                        } else {
                            const float4 t = s_rgb[e];  // (written by this very thread)
                            s_p[sp] = make_float4(E * t.x, E * t.y, E * t.z, E);
                        }
                    }
                }
            }
            __syncthreads();  // P_l complete (and the previous level's pass B is done with s_hs)
            const float(&wl)[kFastRows] = wl_all[l];
            if (!wide) {
                // The box filter in two passes (round 5): the kernel was LDS-bound -- 40 M wave-level LDS instructions per 100
                // frames, 0.5 of its 0.85 ms -- because every thread summed the (2S+1) x (2S+2) window rows of its two outputs
                // itself: 94 16-byte reads per output over the four levels.  Pass A: the window-ROW sums Hs(y, x) = sum_dx
                // P(y, x + dx), once per staged row and output column (dx ascending from 0: the order box_rows used), to LDS;
                // pass B: a thread adds the 2S + 1 row sums of each of its outputs (rows ascending: box_rows' order again).
                // 52 reads per output.  (The paired pass A below adds a window row's inner values in another order than the
                // one-pass form did: this tolerance route's sums are NOT bit-identical to round 4's, they differ by an ulp.)
                constexpr int S = l + 1, HR = kFastH + 2 * S;
                // (pass A, two adjacent output columns 2 xp, 2 xp + 1 per thread: the 2 S + 2 values of their windows are read
                //  once, the 2 S - 1... values both windows hold are added once -- t1 + .. + t2S, ascending -- and each sum gets its
                //  own end: the tolerance route's sums in another order than the one-pass form's, differences of an ulp)
#pragma unroll
                for (int it = 0; it < (HR * (kFastW / 2) + 255) / 256; ++it) {
                    const int idx = tid + it * 256;
                    if (idx < HR * (kFastW / 2)) {
                        const int ya = idx / (kFastW / 2), xp = idx - ya * (kFastW / 2);
                        const float4* row = s_p + (L - S + ya) * SWP + xp;
                        float2v rg = {0.f, 0.f}, bs = {0.f, 0.f};
                        float4 t0, tl;
#pragma unroll
                        for (int k = 0; k <= 2 * S + 1; ++k) {
                            constexpr int c0 = L - S;  // staged column of the first window value of output column 0
                            const float4 t = row[((c0 + k) & 1) * HALF + ((c0 + k) >> 1)];
                            if (k == 0)
                                t0 = t;
                            else if (k == 2 * S + 1)
                                tl = t;
                            else {
                                rg += float2v{t.x, t.y};
                                bs += float2v{t.z, t.w};
                            }
                            // (at most four 16-byte values in flight: the scheduler otherwise hoists all 2 S + 2 reads -- 40 VGPRs
                            //  at S = 4 next to the prefetched tile -- and the packed form spills)
                            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                        }
                        s_hs[ya * SWP + xp] = make_float4(t0.x + rg.x, t0.y + rg.y, t0.z + bs.x, t0.w + bs.y);
                        s_hs[ya * SWP + HALF + xp] = make_float4(rg.x + tl.x, rg.y + tl.y, bs.x + tl.z, bs.y + tl.w);
                    }
                }
                __syncthreads();  // the row sums are complete (and s_p may be rewritten by the next level)
                // (packed adds: the same sums as four scalar adds per value, half the instructions)
                float2v acc01[kFastRows], acc23[kFastRows];
#pragma unroll
                for (int o = 0; o < kFastRows; ++o) acc01[o] = acc23[o] = float2v{0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 2 * S + kFastRows; ++j) {
                    const float4 t = s_hs[(ry * kFastRows + j) * SWP + (lx & 1) * HALF + (lx >> 1)];
#pragma unroll
                    for (int o = 0; o < kFastRows; ++o) {
                        if (j - o >= 0 && j - o <= 2 * S) {
                            acc01[o] += float2v{t.x, t.y};
                            acc23[o] += float2v{t.z, t.w};
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < kFastRows; ++r) {
                    const float ww = div_refined(wl[r], acc23[r].y);  // (the tolerance route: reciprocal + one residual correction, not an IEEE division)
                    o[r][0] += acc01[r].x * ww;
                    o[r][1] += acc01[r].y * ww;
                    o[r][2] += acc23[r].x * ww;
                }
            } else {  // per-pixel maximum, taps from global memory
#pragma unroll
                for (int r = 0; r < kFastRows; ++r) {
                    const float4 c4 = PACKED ? filter_level_wide(reinterpret_cast<const float*>(packed + 4 + l), img_in, l + 1, H, W, px,
                                                                 py0 + r, wl[r], 8)
                                             : filter_level_wide(guidance + l * HW, img_in, l + 1, H, W, px, py0 + r, wl[r], 0);
                    o[r][0] += c4.x;
                    o[r][1] += c4.y;
                    o[r][2] += c4.z;
                }
            }
        };
        level(std::integral_constant<int, 0>{});
        if constexpr (L >= 2) level(std::integral_constant<int, 1>{});
        if constexpr (L >= 3) level(std::integral_constant<int, 2>{});
        if constexpr (L >= 4) level(std::integral_constant<int, 3>{});
        if constexpr (L >= 5) level(std::integral_constant<int, 4>{});
        if constexpr (L >= 6) level(std::integral_constant<int, 5>{});
#pragma unroll
        for (int r = 0; r < kFastRows; ++r)
            if (px < W && py0 + r < H) img_out[(int64_t)(py0 + r) * W + px] = make_float4(o[r][0], o[r][1], o[r][2], 1.0f);
    }
}

// tiles per workgroup of filter_fast: kFastStrip when that still gives every CU its 8 workgroups, else shorter strips
static int fast_strip_for(int tiles_x, int tiles_y, int n) {
    int s = kFastStrip;
    while (s > 1 && (int64_t)((tiles_x + s - 1) / s) * tiles_y * n < 2048) --s;
    return s;
}

hipError_t launch_filter_fast_packed(const void* packed_maps, int H, int W, int n, const float* img_in, float* img_out,
                                     const uint32_t* tile_mask, int mask_words, const float* fill_tile, int sparse, float background,
                                     const uint32_t* fill_maps, hipStream_t stream) {
    if (sparse && (!tile_mask || !fill_maps)) return hipErrorInvalidValue;
    const int tiles_x = (W + kFastW - 1) / kFastW, tiles_y = (H + kFastH - 1) / kFastH, strip = fast_strip_for(tiles_x, tiles_y, n);
    // (grid.x such that ceil(tiles_x / grid.x) == the strip the kernel derives: ceil(tiles_x / strip) workgroups per tile row)
    const dim3 grid((tiles_x + strip - 1) / strip, tiles_y, n), block(256);
    const size_t lds = (size_t)(2 * (kFastH + 8) * kFastRowStride) * sizeof(float4);  // P_l tile + window-row sums, parity-split rows
    FilterCull cull;
    cull.mask = tile_mask;
    cull.mask_words = mask_words;
    cull.tiles_x = (W + 7) / 8;
    cull.fill = reinterpret_cast<const float4*>(fill_tile);
    cull.sparse = sparse ? 1 : 0;
    cull.bg = background;
    cull.fill_maps = sparse ? make_uint4(fill_maps[0], fill_maps[1], fill_maps[2], fill_maps[3]) : make_uint4(0u, 0u, 0u, 0u);
    hipLaunchKernelGGL((filter_fast<4, true>), grid, block, lds, stream, reinterpret_cast<const float*>(packed_maps),
                       (const float*)nullptr, reinterpret_cast<const float4*>(img_in), reinterpret_cast<float4*>(img_out), H, W, cull);
    return hipGetLastError();
}

hipError_t launch_filter_fast(const float* weight, const float* guidance, int L, int H, int W, int n, const float* img_in,
                              float* img_out, hipStream_t stream) {
    return launch_filter_fast_culled(weight, guidance, L, H, W, n, img_in, img_out, nullptr, 0, nullptr, stream);
}

hipError_t launch_filter_fast_culled(const float* weight, const float* guidance, int L, int H, int W, int n, const float* img_in,
                                     float* img_out, const uint32_t* tile_mask, int mask_words, const float* fill_tile,
                                     hipStream_t stream) {
    const FilterCull cull{tile_mask, mask_words, (W + 7) / 8, reinterpret_cast<const float4*>(fill_tile)};
    const int tiles_x = (W + kFastW - 1) / kFastW, tiles_y = (H + kFastH - 1) / kFastH, strip = fast_strip_for(tiles_x, tiles_y, n);
    // (grid.x such that ceil(tiles_x / grid.x) == the strip the kernel derives: ceil(tiles_x / strip) workgroups per tile row)
    const dim3 grid((tiles_x + strip - 1) / strip, tiles_y, n), block(256);
    const float4* in4 = reinterpret_cast<const float4*>(img_in);
    float4* out4 = reinterpret_cast<float4*>(img_out);
#define RTO_FFAST(LL)                                                                                              \
    case LL: {                                                                                                     \
        const size_t lds = (size_t)(2 * (kFastH + 2 * LL) * kFastRowStride + (kFastW + 2 * LL) * (kFastH + 2 * LL)) * sizeof(float4); \
        hipLaunchKernelGGL((filter_fast<LL, false>), grid, block, lds, stream, weight, guidance, in4, out4, H, W, cull); \
    } break;
    switch (L) {
        RTO_FFAST(1) RTO_FFAST(2) RTO_FFAST(3) RTO_FFAST(4) RTO_FFAST(5) RTO_FFAST(6)
        default: return hipErrorInvalidValue;
    }
#undef RTO_FFAST
    return hipGetLastError();
}

static hipError_t launch_filter_impl(const float* weight, const float* guidance, int L, int H, int W, int n, const float* img_in,
                                     float* img_out, float* rgb_filtered, float* max_map, float* inv_kernel_sum,
                                     const FilterCull& cull, hipStream_t stream);

hipError_t launch_filter(const float* weight, const float* guidance, int L, int H, int W, int n, const float* img_in,
                         float* img_out, hipStream_t stream) {
    return launch_filter_impl(weight, guidance, L, H, W, n, img_in, img_out, nullptr, nullptr, nullptr, FilterCull{nullptr, 0, 0, nullptr, 0, 0.f, make_uint4(0u, 0u, 0u, 0u)}, stream);
}

hipError_t launch_filter_culled(const float* weight, const float* guidance, int L, int H, int W, int n, const float* img_in,
                                float* img_out, const uint32_t* tile_mask, int mask_words, const float* fill_tile, hipStream_t stream) {
    return launch_filter_impl(weight, guidance, L, H, W, n, img_in, img_out, nullptr, nullptr, nullptr,
                              FilterCull{tile_mask, mask_words, (W + 7) / 8, reinterpret_cast<const float4*>(fill_tile), 0, 0.f, make_uint4(0u, 0u, 0u, 0u)}, stream);
}

// rgb_filtered == nullptr: inference forward; else all three save arrays are written too
hipError_t launch_filter_train(const float* weight, const float* guidance, int L, int H, int W, int n,
                               const float* img_in, float* img_out, float* rgb_filtered, float* max_map,
                               float* inv_kernel_sum, hipStream_t stream) {
    return launch_filter_impl(weight, guidance, L, H, W, n, img_in, img_out, rgb_filtered, max_map, inv_kernel_sum,
                              FilterCull{nullptr, 0, 0, nullptr, 0, 0.f, make_uint4(0u, 0u, 0u, 0u)}, stream);
}

static hipError_t launch_filter_impl(const float* weight, const float* guidance, int L, int H, int W, int n, const float* img_in,
                                     float* img_out, float* rgb_filtered, float* max_map, float* inv_kernel_sum,
                                     const FilterCull& cull, hipStream_t stream) {
    const dim3 grid((W + kFiltW - 1) / kFiltW, (H + kFiltH - 1) / kFiltH, n), block(256);
    const float4* in4 = reinterpret_cast<const float4*>(img_in);
    float4* out4 = reinterpret_cast<float4*>(img_out);
    float4* rf4 = reinterpret_cast<float4*>(rgb_filtered);
#define RTO_FILT(LL)                                                                                                   \
    case LL:                                                                                                           \
        if (rgb_filtered)                                                                                              \
            hipLaunchKernelGGL((filter_fused<LL, true>), grid, block, 0, stream, weight, guidance, in4, out4, H, W, rf4, \
                               max_map, inv_kernel_sum, cull);                                                         \
        else                                                                                                           \
            hipLaunchKernelGGL((filter_fused<LL, false>), grid, block, 0, stream, weight, guidance, in4, out4, H, W,    \
                               (float4*)nullptr, (float*)nullptr, (float*)nullptr, cull);                              \
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
