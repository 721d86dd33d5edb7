// guidance_kernels.hip -- the compact GuidanceNet as ONE fused gfx950 kernel.
//
// Reference network (denoiser/network.py:123-168 after compact_and_compile :170-208, run by
// renderer/src/denoiser/denoiser.cpp:46):  x = aux.half();  x = relu6(conv3x3(x; 8 -> C1));
// x = relu6(conv3x3(x; C1 -> 2L));  x = x.float();  weight = softmax(x[:L]);  guidance = x[L:].
// Default C1 = 32, L = 4 (denoiser/configs/blender.txt:21-25): 5.9 GFLOP per 800x800 frame, which
// PyTorch-ROCm runs as two MIOpen Winograd convolutions plus seven elementwise launches
// (~0.2 ms/frame, profiles/r1_b_*).  Here one workgroup produces a 32x8 pixel tile end to end:
//
//   stage A  aux tile + 2-pixel halo, fp32 planar -> fp16 HWC in LDS (zero outside the image);
//            one thread per tile pixel, its 8 channel loads in flight together
//   stage B  layer 1 on the tile + 1-pixel halo with v_mfma_f32_16x16x32_f16: M = 16 output
//            channels (weights, A operand, register-resident), N = 16 pixels (B operand: ONE
//            ds_read_b128 per lane = the 8 input channels of one tap), K = 9 taps x 8 channels
//            padded to 96; + bias, ReLU6, fp16, HWC in LDS (zero outside the image = the second
//            convolution's "same" padding)
//   stage C  layer 2 the same way: K = 9 taps x 32 channels = 9 MFMAs per 16 pixels, output
//            channels padded 8 -> 16; + bias, ReLU6, fp16 rounding (the reference keeps fp16
//            activations), then softmax over the L weight channels in fp32 and the stores
//
// Accumulation is fp32 inside the MFMA like cuDNN/MIOpen fp16 convolutions; rounding points
// (fp16 input, fp16 activations after each ReLU6) are the reference's.  Results agree with the fp32
// PyTorch network to fp16 accuracy (tests/test_guidance_fused.py), not bit for bit: libtorch 1.11 /
// cuDNN results are not pinned by the reference either.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "rto_launch.h"

namespace rto {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int kGW = 32, kGH = 8;  // output tile
#ifndef RTO_NET_STRIP
#define RTO_NET_STRIP 1
#endif
constexpr int kStrip = RTO_NET_STRIP;  // tiles per workgroup, along x
constexpr int kCIn = 8;           // aux channels (render_context.hpp:23)

__device__ __forceinline__ float relu6(float x) { return fminf(fmaxf(x, 0.f), 6.f); }

// C1 = mid channels (multiple of 16, <= 64), L = kernel levels (2L <= 16)
// SQ: the aux planes 4..7 are the squares of planes 0..3 (what the renderer writes, volrend.cu:195-202:
// a[4 + c] = out[c] * out[c], one fp32 multiply): read only planes 0..3 and square them here -- the same fp32
// product, so the same fp16 inputs, from half the bytes.
// PACK: write the network's 8 output channels as they leave the last ReLU6 -- fp16 values, [n][H][W][8]: the 4
// softmax logits, then the 4 guidance values -- instead of fp32 weight_map / guidance_map planes.  16 B per pixel
// instead of 32, nothing lost: the reference's `.float()` (network.py:112) only widens those fp16 values, and the
// consumer (filter_fast<L, true>) applies softmax_weights() below to the logits itself.
template <int C1, int L, bool SQ, bool PACK>
__global__ void __launch_bounds__(256) guidance_fused(const float* __restrict__ aux,    // [n][8][H][W]
                                                       const _Float16* __restrict__ w1,  // [C1][96]   k = tap*8 + ci
                                                       const float* __restrict__ b1,     // [C1]
                                                       const _Float16* __restrict__ w2,  // [16][9*C1] k = tap*C1 + ci
                                                       const float* __restrict__ b2,     // [16]
                                                       float* __restrict__ weight_out,   // [n][L][H][W]
                                                       float* __restrict__ guidance_out, // [n][L][H][W]
                                                       int H, int W) {
    constexpr int IW = kGW + 4, IH = kGH + 4;  // input tile with halo 2
    constexpr int AW = kGW + 2, AH = kGH + 2;  // layer-1 activation tile with halo 1
    constexpr int NT1 = C1 / 16;               // output-channel tiles of layer 1
    constexpr int KS2 = 9 * C1 / 32;           // k-steps of layer 2
    // activation pixel stride in halves: C1 + 8, i.e. 80 B instead of 64 B for C1 = 32 -- with a
    // 64-byte stride the 16 pixels a ds_read_b128 gathers start on only two distinct bank groups
    // (8-way conflict); at 80 B every bank is touched exactly twice, the minimum for 256 B
#ifndef RTO_NET_PAD
#define RTO_NET_PAD 0
#endif
    constexpr int AS = C1 + RTO_NET_PAD;
    __shared__ __attribute__((aligned(16))) _Float16 s_in[IH * IW * kCIn];
    __shared__ __attribute__((aligned(16))) _Float16 s_act[AH * AW * AS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // A workgroup can walk a strip of kStrip tiles along x with the input of tile t + 1 on its way into registers
    // while layers 1 and 2 of tile t run.  Measured (16 frames of 800x800): strip 1 0.356 ms, 3 0.376-0.44, 5 0.39-0.44,
    // 25 0.47 -- with 4 workgroups per CU the other workgroups already cover a tile's load phase, and longer strips
    // only leave fewer workgroups to balance; the default stays 1.
    const int tiles_x = (W + kGW - 1) / kGW;
    const int tx_first = blockIdx.x * kStrip;
    const int y0 = blockIdx.y * kGH;
    const int64_t HW = (int64_t)H * W;
    aux += (int64_t)blockIdx.z * kCIn * HW;
    weight_out += (int64_t)blockIdx.z * L * HW;  // (PACK: [H][W][8] fp16 = 4 floats per pixel = L * HW floats per image too)
    guidance_out += (int64_t)blockIdx.z * L * HW;

    const int col = lane & 15, kg = lane >> 4;  // MFMA lane roles: pixel (B/C column), k-group / row block

    // Weights first: their loads overlap stage A instead of stalling the first MFMAs of each layer.
    half8 wa[NT1][3];
#pragma unroll
    for (int t = 0; t < NT1; ++t)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
            wa[t][ks] = *reinterpret_cast<const half8*>(w1 + (size_t)(t * 16 + col) * 96 + ks * 32 + kg * 8);
    half8 wb[KS2];
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks)
        wb[ks] = *reinterpret_cast<const half8*>(w2 + (size_t)col * (9 * C1) + ks * 32 + kg * 8);

    // biases once per workgroup, before the strip loop: a global load inside stages B / C would make their
    // s_waitcnt drain the prefetch of the next tile as well (vmcnt retires in issue order)
    float bias1[NT1][4], bias2[4];
#pragma unroll
    for (int t = 0; t < NT1; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias1[t][i] = b1[t * 16 + kg * 4 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias2[i] = b2[kg * 4 + i];

    // ---- stage A: input tile, planar fp32 -> HWC fp16.  One thread = one tile pixel: its channel loads are
    // independent (all in flight at once) and become one 16-byte LDS store.
    constexpr int NPIX = IH * IW, NIT = (NPIX + 255) / 256;
    constexpr int NLD = SQ ? kCIn / 2 : kCIn;  // planes actually read
    float v[NIT][NLD];
    auto fetch = [&](int x0) {  // issue the loads of the tile whose first output column is x0
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * 256;
            const int ty = e / IW, tx = e - ty * IW;
            const int gx = x0 - 2 + tx, gy = y0 - 2 + ty;
            const bool in = e < NPIX && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const int64_t gi = in ? (int64_t)gy * W + gx : 0;
#pragma unroll
            for (int c = 0; c < NLD; ++c) {
                const float t = aux[c * HW + gi];
                v[it][c] = in ? t : 0.f;
            }
        }
    };
#ifdef RTO_NET_DBG_STAMP
    unsigned long long st[8];
    st[0] = __builtin_amdgcn_s_memtime();
#endif
    fetch(tx_first * kGW);

    for (int ts = 0; ts < kStrip; ++ts) {
    const int tile_x = tx_first + ts;
    if (tile_x >= tiles_x) break;  // workgroup-uniform
    const int x0 = tile_x * kGW;
    const bool tile_interior = x0 >= 1 && x0 + kGW + 1 <= W && y0 >= 1 && y0 + kGH + 1 <= H;  // layer-1 region inside the image
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * 256;
        if (e < NPIX) {
            half8 h;
#pragma unroll
            for (int c = 0; c < NLD; ++c) h[c] = (_Float16)v[it][c];
            if (SQ) {
#pragma unroll
                for (int c = 0; c < kCIn / 2; ++c) h[kCIn / 2 + c] = (_Float16)(v[it][c] * v[it][c]);
            }
            *reinterpret_cast<half8*>(s_in + (size_t)e * kCIn) = h;
        }
    }
#ifdef RTO_NET_DBG_STAMP
    if (ts == 0) st[1] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();  // s_in complete; every wave is also done with stage C of the previous tile (s_act is free)
#ifdef RTO_NET_DBG_STAMP
    if (ts == 0) st[2] = __builtin_amdgcn_s_memtime();
#endif
    if (ts + 1 < kStrip && tile_x + 1 < tiles_x) fetch(x0 + kGW);  // next tile's input: in flight during stages B and C


    // ---- stage B: layer 1 on the AH x AW region
    {
        const float(&bias)[NT1][4] = bias1;
        constexpr int NG1 = (AH * AW + 15) / 16;
#ifndef RTO_NET_DBG_BREP
#define RTO_NET_DBG_BREP 1
#endif
        for (int rep = 0; rep < RTO_NET_DBG_BREP; ++rep)
        for (int g = wave; g < NG1; g += 4) {
            if (RTO_NET_DBG_BREP > 1) asm volatile("" ::: "memory");
            // (the index arithmetic of this loop is what the kernel's VALU time went into: 24-bit multiplies and
            // a reciprocal multiply instead of 32-bit mul_lo / mul_hi -- all operands are far below 2^24)
            const int p = g * 16 + col;
            const bool valid = p < AH * AW;
            static_assert(AH * AW < 2048, "reciprocal division below assumes a small tile");
            const int q = (int)(__umul24((unsigned)p, (65536u + AW - 1) / AW) >> 16);  // p / AW for p < 2048
            const int ry = valid ? q : 0, rx = valid ? p - (int)__umul24((unsigned)q, AW) : 0;
            // the accumulators start from the bias (lane element i of tile t is output channel t*16 + kg*4 + i): one
            // operand of the first MFMA instead of eight additions per group
            float4v acc[NT1];
#pragma unroll
            for (int t = 0; t < NT1; ++t) acc[t] = (float4v){bias[t][0], bias[t][1], bias[t][2], bias[t][3]};
            half8 bf1[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int tap = ks * 4 + kg;
                bf1[ks] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
                if (tap < 9) {
                    const int ky = tap / 3, kx = tap - ky * 3;
                    bf1[ks] = *reinterpret_cast<const half8*>(s_in + (__umul24((unsigned)(ry + ky), IW) + rx + kx) * kCIn);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int t = 0; t < NT1; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][ks], bf1[ks], acc[t], 0, 0, 0);
            if (valid) {
                // outside the image the activation is the second convolution's zero padding.  Most tiles lie inside with
                // their halo (workgroup-uniform test): no mask at all there; border tiles scale by 0 or 1 (relu6(...) is
                // finite and >= 0, so x * 1 = x and x * 0 = +0 exactly)
                float inside = 1.f;
                if (!tile_interior) {
                    const int gx = x0 - 1 + rx, gy = y0 - 1 + ry;
                    inside = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? 1.f : 0.f;
                }
#pragma unroll
                for (int t = 0; t < NT1; ++t) {
                    half4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float r = __builtin_amdgcn_fmed3f(acc[t][i], 0.f, 6.f);  // ReLU6
                        o[i] = (_Float16)(tile_interior ? r : r * inside);
                    }
                    *reinterpret_cast<half4*>(s_act + __umul24((unsigned)p, AS) + t * 16 + kg * 4) = o;
                }
            }
        }
    }
#ifdef RTO_NET_DBG_STAMP
    if (ts == 0) st[3] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef RTO_NET_DBG_STAMP
    if (ts == 0) st[4] = __builtin_amdgcn_s_memtime();
#endif

    // ---- stage C: layer 2 on the kGH x kGW tile, softmax, stores
    {
        const float(&bias)[4] = bias2;
        constexpr int NG2 = kGH * kGW / 16;
#ifndef RTO_NET_DBG_CREP
#define RTO_NET_DBG_CREP 1
#endif
        for (int rep = 0; rep < RTO_NET_DBG_CREP; ++rep)
        for (int g = wave; g < NG2; g += 4) {
            if (RTO_NET_DBG_CREP > 1) asm volatile("" ::: "memory");
            const int p = g * 16 + col;
            const int oy = p / kGW, ox = p - oy * kGW;
            float4v acc = (float4v){bias[0], bias[1], bias[2], bias[3]};  // the bias: the first MFMA's C operand
            // all of the group's B fragments first, each into registers of its own, then the MFMA chain: with one
            // fragment register set the compiler serialises read -> wait -> MFMA nine times and the group costs nine
            // LDS latencies (measured: 1.5 k clocks per group at 4 waves per SIMD)
            half8 bf[KS2];
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                const int k0 = ks * 32 + kg * 8;  // k = tap*C1 + ci
                const int tap = k0 / C1, ci = k0 - tap * C1;
                const int ky = tap / 3, kx = tap - ky * 3;
                bf[ks] = *reinterpret_cast<const half8*>(s_act + __umul24(__umul24((unsigned)(oy + ky), AW) + ox + kx, AS) + ci);
            }
            __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the reads back between the MFMAs)
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ks], bf[ks], acc, 0, 0, 0);
            const int gx = x0 + ox, gy = y0 + oy;
            if (gx < W && gy < H && kg * 4 < 2 * L) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (float)(_Float16)__builtin_amdgcn_fmed3f(acc[i], 0.f, 6.f);  // ReLU6 -> fp16 activations, then .float()
                const int64_t pix = (int64_t)gy * W + gx;
                if (PACK) {  // (weight_out = the packed buffer of image blockIdx.z; kg 0: logits, kg 1: guidance)
                    half4 h;
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[i] = (_Float16)v[i];
                    *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(weight_out) + pix * 8 + kg * 4) = h;
                } else if (L == 4) {
                    if (kg == 0) {  // channels 0..3: softmax -> weight_map (network.py:113-114)
                        float wgt[4];
                        softmax_weights4(v, wgt);
#pragma unroll
                        for (int i = 0; i < 4; ++i) weight_out[i * HW + pix] = wgt[i];
                    } else {  // channels 4..7: guidance_map (:116)
#pragma unroll
                        for (int i = 0; i < 4; ++i) guidance_out[i * HW + pix] = v[i];
                    }
                }
            }
        }
    }
#ifdef RTO_NET_DBG_STAMP
    if (ts == 0) {
        st[5] = __builtin_amdgcn_s_memtime();
        if ((blockIdx.x == 3 || blockIdx.x == 11) && (blockIdx.y == 40 || blockIdx.y == 41) && blockIdx.z == 2 && lane == 0)
            printf("blk %d,%d wave %d: fetch+wait+cvt %llu | sync %llu | B %llu | sync %llu | C %llu | start %llu\n", blockIdx.x, blockIdx.y, wave,
                   st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[0]);
    }
#endif
    }  // strip
}

}  // namespace

hipError_t launch_guidance_net(const float* aux, const void* w1, const float* b1, const void* w2, const float* b2, int c1,
                               int levels, int n, int H, int W, float* weight_out, float* guidance_out,
                               bool squares_implied, hipStream_t stream) {
    if (c1 != 32 || levels != 4) return hipErrorInvalidValue;  // the reference configuration (blender.txt:21-25)
    const int tiles_x = (W + kGW - 1) / kGW;
    const dim3 grid((tiles_x + kStrip - 1) / kStrip, (H + kGH - 1) / kGH, n), block(256);
    const bool pack = guidance_out == nullptr;  // weight_out is then the packed fp16 buffer [n][H][W][8]
#define RTO_NET(SQ, PK)                                                                                                   \
    hipLaunchKernelGGL((guidance_fused<32, 4, SQ, PK>), grid, block, 0, stream, aux, (const _Float16*)w1, b1, (const _Float16*)w2, \
                       b2, weight_out, guidance_out, H, W)
    if (pack) {
        if (squares_implied) RTO_NET(true, true); else RTO_NET(false, true);
    } else {
        if (squares_implied) RTO_NET(true, false); else RTO_NET(false, false);
    }
#undef RTO_NET
    return hipGetLastError();
}

}  // namespace rto
