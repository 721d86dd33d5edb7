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
//            channels padded 8 -> 16, each activation-row fragment shared by the three output rows
//            it serves; + bias, ReLU6, fp16 rounding (the reference keeps fp16 activations), then
//            softmax over the L weight channels in fp32 and the stores
//
// Accumulation is fp32 inside the MFMA like cuDNN/MIOpen fp16 convolutions; rounding points
// (fp16 input, fp16 activations after each ReLU6) are the reference's.  Results agree with the fp32
// PyTorch network to fp16 accuracy (tests/test_guidance_fused.py), not bit for bit: libtorch 1.11 /
// cuDNN results are not pinned by the reference either.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <type_traits>

#include "rto_launch.h"
#include "rto_denoise_launch.h"

namespace rto {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int kGW = 32, kGH = 8;  // output tile
// Tiles per workgroup, along x.  Round 2 chose 5 when every tile was computed (strips of 1 / 2 / 5 / 7..25: 0.70 / 0.64 / 0.60 /
// 0.59-0.62 ms per 50 frames).  With the culling two thirds of the workgroups only test their tiles' marks and leave, and each of
// those costs a launch and a dependent load: round 6 re-measured 5 / 7 / 9 / 13 tiles per workgroup at 0.686 / 0.862 / 0.626 /
// 0.616 ms per 100 C2 frames (7 splits a 25-tile row 7 + 7 + 7 + 4) and 1.86 / - / 1.45 / 1.38 ms on C4 (60-tile rows):
// 13 = the most that the one-thread-per-(tile, render tile) mark test below fits into 256 threads (profiles/r6_m_ab_strips*.txt)
#ifndef RTO_NET_STRIP
#define RTO_NET_STRIP 13
#endif
constexpr int kStrip = RTO_NET_STRIP;
constexpr int kCIn = 8;           // aux channels (render_context.hpp:23)

__device__ __forceinline__ float relu6(float x) { return fminf(fmaxf(x, 0.f), 6.f); }

// Empty-space culling carried into the network (round 3; PACK route): the render context's tile marks (FrameBatch::tile_mask)
// and the network's output for a pixel whose 5x5 aux neighbourhood is background -- 8 fp16 values, measured once by this
// kernel on a synthetic background frame.  mask == nullptr: off.
struct NetCull {
    const uint32_t* mask;
    int mask_words, tiles_x;
    uint4 fill;      // PACK: the 8 fp16 outputs of a pixel whose 5x5 aux neighbourhood is background
    float planes[8]; // otherwise: the same as 4 softmax weights + 4 guidance values
    // SPARSE lean frames (round 6, rto_ctx_set_lean_outputs level 2; IN = 2, PACK): the renderer stored no pixel of an unmarked
    // (culled) render tile -- such a pixel IS the background (bg, bg, bg, alpha 0), so the staging loop substitutes it -- and
    // this kernel stores no maps for the tiles it skips: the filter substitutes `fill` for them the same way
    int sparse;
    float bg;
};

// C1 = mid channels (multiple of 16, <= 64), L = kernel levels (2L <= 16)
// SQ: the aux planes 4..7 are the squares of planes 0..3 (what the renderer writes, volrend.cu:195-202:
// a[4 + c] = out[c] * out[c], one fp32 multiply): read only planes 0..3 and square them here -- the same fp32
// product, so the same fp16 inputs, from half the bytes.
// PACK: write the network's 8 output channels as they leave the last ReLU6 -- fp16 values, [n][H][W][8]: the 4
// softmax logits, then the 4 guidance values -- instead of fp32 weight_map / guidance_map planes.  16 B per pixel
// instead of 32, nothing lost: the reference's `.float()` (network.py:112) only widens those fp16 values, and the
// consumer (filter_fast<L, true>) applies softmax_weights() below to the logits itself.
// Launch bounds per instantiation: with all 8 aux planes in flight (SQ = false) the kernel needs 134 VGPRs; bounded to 128
// (4 workgroups per CU) it spilled 2-4 of them to scratch and ran slower.  It was also the kernel in whose company the
// bit-exact filter first lost its determinism on a shared GPU (DESIGN_HISTORY.md "Determinism when the GPU is shared": the cause
// turned out to be v_pk_fma_f32 under MFMA load from other processes, this kernel being the longest MFMA kernel around); no
// kernel of the denoise stage uses scratch now (tests/test_codegen.py).  -DRTO_NET_SQ0_WG=4 rebuilds the old bound.
#ifndef RTO_NET_SQ0_WG
#define RTO_NET_SQ0_WG 3
#endif
// IN (round 5): 0 = all 8 aux planes, 1 = SQ, 2 = SQ from the INTERLEAVED image a lean batched launch leaves ([n][H][W][4] =
// r, g, b, alpha -- the values of aux planes 0..3, rto_ctx_set_lean_outputs): one 16-byte load per staged pixel.
template <int C1, int L, int IN, bool PACK>
__global__ void __launch_bounds__(256, IN ? 4 : RTO_NET_SQ0_WG) guidance_fused(const float* __restrict__ aux,    // [n][8][H][W] (IN = 2: [n][H][W][4])
                                                       const _Float16* __restrict__ w1,  // [C1][96]   k = tap*8 + ci; k = 72: bias
                                                       const _Float16* __restrict__ w2,  // [16][9*C1] k = tap*C1 + ci
                                                       const float* __restrict__ b2,     // [16]
                                                       float* __restrict__ weight_out,   // [n][L][H][W]
                                                       float* __restrict__ guidance_out, // [n][L][H][W]
                                                       int H, int W, const NetCull cull, const int strip /* tiles per workgroup, <= kStrip */) {
    constexpr int IW = kGW + 4, IH = kGH + 4;  // input tile with halo 2
    constexpr int AW = kGW + 2, AH = kGH + 2;  // layer-1 activation tile with halo 1
    constexpr int NT1 = C1 / 16;               // output-channel tiles of layer 1
    constexpr int KS2 = 9 * C1 / 32;           // k-steps of layer 2
    // activation pixel stride in halves: C1 + 8, i.e. 80 B instead of 64 B for C1 = 32 -- with a
    // 64-byte stride the 16 pixels a ds_read_b128 gathers start on only two distinct bank groups
    // (8-way conflict); at 80 B every bank is touched exactly twice, the minimum for 256 B
#ifndef RTO_NET_PAD
#define RTO_NET_PAD 8
#endif
    constexpr int AS = C1 + RTO_NET_PAD;
    __shared__ __attribute__((aligned(16))) _Float16 s_in[IH * IW * kCIn];
    __shared__ __attribute__((aligned(16))) _Float16 s_act[AH * AW * AS];
    __shared__ __attribute__((aligned(16))) _Float16 s_pad[16];  // B fragments of the padding taps: {1,0,..,0} (bias slot), {0,..,0}
    __shared__ __attribute__((aligned(16))) float s_b2[16];      // layer-2 bias

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 16) {  // (published by the barrier after stage A)
        s_pad[tid] = (_Float16)(tid == 0 ? 1.f : 0.f);
        s_b2[tid] = b2[tid];
    }
    // A workgroup walks a strip of kStrip tiles along x with the input of tile t + 1 on its way into registers while
    // layers 1 and 2 of tile t run (a tile's input fetch is 37 % of its time when nothing hides it).  Measured per 50
    // frames of 800x800 once the stages had been slimmed down to fit the prefetch registers without spills: strip 1
    // 0.70 ms, 2 0.64, 5 0.60, 7..25 0.59-0.62.  (While stage B still spent ~75 VALU instructions per group the kernel
    // was issue-bound and strips gained nothing.)
    const int tiles_x = (W + kGW - 1) / kGW;
    const int tx_first = blockIdx.x * strip;
    const int y0 = blockIdx.y * kGH;
    const int64_t HW = (int64_t)H * W;
    constexpr bool SQ = IN != 0;
    aux += (int64_t)blockIdx.z * (IN == 2 ? 4 : kCIn) * HW;
    weight_out += (int64_t)blockIdx.z * L * HW;  // (PACK: [H][W][8] fp16 = 4 floats per pixel = L * HW floats per image too)
    guidance_out += (int64_t)blockIdx.z * L * HW;

    const int col = lane & 15, kg = lane >> 4;  // MFMA lane roles: pixel (B/C column), k-group / row block

    // Tiles of the strip whose input region (tile + halo 2) lies inside the image and in culled render tiles read nothing
    // but background: every output pixel is cull.fill.  Bit ts of `skip_tiles`; workgroup-uniform.
    // Round 6: every wave builds the marks of the render tiles under the strip's input regions for itself -- 3 tile rows from
    // (y0 - 2) >> 3, 64 columns from 4 tx_first - 1 (a tile's region spans 6 of them from bit 4 ts), lane = column, the three mark
    // words of a lane requested together, the rows by ballot: one memory round trip, no LDS, no barrier (the first form tested one
    // (tile, render tile) per thread into LDS words under two barriers).  A workgroup with nothing to compute leaves before it
    // has requested a single weight.
    uint32_t skip_tiles = 0;
    constexpr int RX = (IW + 7) / 8 + 1, RY = (IH + 7) / 8 + 1;  // render tiles an input region can touch, per axis
    static_assert(RX == 6 && RY == 3 && 4 * (kStrip - 1) + RX <= 64, "64 columns x 3 rows of render tiles cover a strip's input regions");
    unsigned long long rmk[RY] = {0ull, 0ull, 0ull};  // bit c of rmk[r]: render tile (4 tx_first - 1 + c, ((y0 - 2) >> 3) + r) is marked
    if (cull.mask) {
        const uint32_t* const fm = cull.mask + (size_t)blockIdx.z * cull.mask_words;
        const uint32_t keep_all = fm[cull.mask_words - 1] & 1u;
        uint32_t word[RY];
        bool inside[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const int tx = 4 * tx_first - 1 + lane, ty = ((y0 - 2) >> 3) + r;
            inside[r] = tx >= 0 && ty >= 0 && tx < cull.tiles_x && ty * 8 < H;
            const uint32_t t = inside[r] ? (uint32_t)(ty * cull.tiles_x + tx) : 0u;
            word[r] = fm[t >> 5] >> (t & 31u);
        }
#pragma unroll
        for (int r = 0; r < RY; ++r) rmk[r] = __builtin_amdgcn_ballot_w64(inside[r] && ((word[r] | keep_all) & 1u) != 0u);
        const unsigned long long any_row = rmk[0] | rmk[1] | rmk[2];
        for (int ts = 0; ts < strip && tx_first + ts < tiles_x; ++ts) {
            const int x0 = (tx_first + ts) * kGW - 2, ry0 = y0 - 2;
            // (outside the image the zero padding is not background: computed)
            if (x0 >= 0 && ry0 >= 0 && x0 + IW <= W && ry0 + IH <= H && ((any_row >> (4 * ts)) & 0x3full) == 0ull) skip_tiles |= 1u << ts;
        }
    }
    // sparse input: bit sy * RX + sx = render tile (sx, sy) of strip tile ts's input region is marked (its pixels were stored)
    auto region_marks = [&](int ts) {
        return (uint32_t)((rmk[0] >> (4 * ts)) & 0x3full) | (uint32_t)((rmk[1] >> (4 * ts)) & 0x3full) << RX |
               (uint32_t)((rmk[2] >> (4 * ts)) & 0x3full) << (2 * RX);
    };
    auto next_live = [&](int ts) {  // first tile >= ts of the strip that has to be computed (kStrip: none)
        while (ts < strip && tx_first + ts < tiles_x && ((skip_tiles >> ts) & 1u)) ++ts;
        return (ts < strip && tx_first + ts < tiles_x) ? ts : kStrip;
    };
    if (PACK && cull.sparse && next_live(0) == kStrip) return;  // (sparse maps: nothing is stored for a skipped tile either)

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

    // Biases cost no registers: layer 1's rides in the weights (k-slot 72, the first padding tap, whose B element is
    // the constant 1 of s_pad; the packer rounds it to fp16 like the reference's `.half()` does), layer 2's is read from
    // LDS once per tile.

    // ---- stage A: input tile, planar fp32 -> HWC fp16.  One thread = one tile pixel: its channel loads are
    // independent (all in flight at once) and become one 16-byte LDS store.
    constexpr int NPIX = IH * IW, NIT = (NPIX + 255) / 256;
    constexpr int NLD = SQ ? kCIn / 2 : kCIn;  // planes actually read
    float v[NIT][NLD];
    auto fetch = [&](int x0) {  // issue the loads of the tile whose first output column is x0
        // (sparse: the marks of the render tiles under this tile's input region, in an SGPR -- whether a pixel was stored is known
        //  before its load is issued, which is then simply not issued: no dependent mask load, no select that waits for the pixel)
        const uint32_t rtm = cull.sparse ? region_marks(x0 / kGW - tx_first) : 0u;
        const int rtx0 = (x0 - 2) >> 3, rty0 = (y0 - 2) >> 3;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * 256;
            const int ty = e / IW, tx = e - ty * IW;
            const int gx = x0 - 2 + tx, gy = y0 - 2 + ty;
            const bool in = e < NPIX && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const int gi = in ? gy * W + gx : 0;  // (8 * H * W < 2^31: rto_ctx_create's size check)
            if constexpr (IN == 2) {
                // (sparse: a pixel of an unmarked render tile was never stored -- it is the background)
                const bool stored = !cull.sparse || ((rtm >> (((gy >> 3) - rty0) * RX + ((gx >> 3) - rtx0))) & 1u) != 0u;
                const float bgv = in && cull.sparse ? cull.bg : 0.f;
                float4 t = make_float4(bgv, bgv, bgv, 0.f);
                if (in && stored) t = reinterpret_cast<const float4*>(aux)[gi];
                v[it][0] = t.x;
                v[it][1] = t.y;
                v[it][2] = t.z;
                v[it][3] = t.w;
            } else {
#pragma unroll
                for (int c = 0; c < NLD; ++c) {
                    const float t = aux[c * (int)HW + gi];
                    v[it][c] = in ? t : 0.f;
                }
            }
        }
    };
#ifdef RTO_NET_DBG_STAMP
    unsigned long long st[8];
    st[0] = __builtin_amdgcn_s_memtime();
#endif
    if (cull.mask && !(PACK && cull.sparse)) {  // the skipped tiles first (interior tiles: every pixel of them is inside the image)
        for (int ts = 0; ts < strip && tx_first + ts < tiles_x; ++ts)
            if ((skip_tiles >> ts) & 1u) {
                const int64_t pix = (int64_t)(y0 + (tid >> 5)) * W + (tx_first + ts) * kGW + (tid & 31);
                if (PACK) {
                    reinterpret_cast<uint4*>(weight_out)[pix] = cull.fill;
                } else {
#pragma unroll
                    for (int l = 0; l < L; ++l) {
                        weight_out[l * HW + pix] = cull.planes[l];
                        guidance_out[l * HW + pix] = cull.planes[4 + l];
                    }
                }
            }
    }
    int ts_live = next_live(0);
    if (ts_live < kStrip) fetch((tx_first + ts_live) * kGW);

#pragma nounroll
    while (ts_live < kStrip) {  // (workgroup-uniform)
    const int x0 = (tx_first + ts_live) * kGW;
    ts_live = next_live(ts_live + 1);
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
    if (x0 == tx_first * kGW) st[1] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();  // s_in complete; every wave is also done with stage C of the previous tile (s_act is free)
#ifdef RTO_NET_DBG_STAMP
    if (x0 == tx_first * kGW) st[2] = __builtin_amdgcn_s_memtime();
#endif
    if (ts_live < kStrip) fetch((tx_first + ts_live) * kGW);  // next computed tile's input: in flight during stages B and C


    // ---- stage B: layer 1 on the AH x AW region
    // The kernel is VALU-bound (stamps: stage B 38 % of a tile's time at ~75 VALU instructions per 16-pixel group
    // around 6 MFMAs), so the loop carries its indices instead of recomputing them: group g + 4 of a wave is 64
    // pixels = one row + 30 columns further on; a tap's LDS address is the group's base plus a per-lane constant; the
    // three padding taps of the last k-step read a zeroed 16-byte slot; tiles whose activation region lies inside the
    // image (all but the border ring) skip the zero-padding mask altogether.
    {
        constexpr int NG1 = (AH * AW + 15) / 16;
        static_assert(AW > 30 && AW <= 64, "the row / column carry below assumes one wrap per 64 pixels");
#ifndef RTO_NET_DBG_BREP
#define RTO_NET_DBG_BREP 1
#endif
        uint32_t tapoff[2];  // byte offset of this lane's tap in k-steps 0 and 1, relative to the group's base pixel
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int tap = ks * 4 + kg, ky = tap / 3, kx = tap - ky * 3;
            tapoff[ks] = (uint32_t)((ky * IW + kx) * kCIn * 2);
        }
        constexpr uint32_t kTap8Off = (uint32_t)((2 * IW + 2) * kCIn * 2);
        const bool tap2_real = kg == 0;  // k-step 2 holds tap 8, the bias slot (kg 1) and two padding taps
        const char* tap2_pad = reinterpret_cast<const char*>(s_pad) + (kg == 1 ? 0 : 16);
        auto layer1 = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
            for (int rep = 0; rep < RTO_NET_DBG_BREP; ++rep) {
                if (RTO_NET_DBG_BREP > 1) asm volatile("" ::: "memory");
                int p = wave * 16 + col;
                int ry = p >= AW ? 1 : 0, rx = p - ry * AW;
                for (int g = wave; g < NG1; g += 4) {
                    const uint32_t base = (uint32_t)((ry * IW + rx) * kCIn * 2);  // bytes into s_in
                    const char* sin_b = reinterpret_cast<const char*>(s_in);
                    half8 bf1[3];
                    bf1[0] = *reinterpret_cast<const half8*>(sin_b + base + tapoff[0]);
                    bf1[1] = *reinterpret_cast<const half8*>(sin_b + base + tapoff[1]);
                    bf1[2] = *reinterpret_cast<const half8*>(tap2_real ? sin_b + base + kTap8Off : tap2_pad);
                    float4v acc[NT1];
#pragma unroll
                    for (int t = 0; t < NT1; ++t) acc[t] = (float4v){0.f, 0.f, 0.f, 0.f};  // (the bias is one of the products)
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                        for (int t = 0; t < NT1; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][ks], bf1[ks], acc[t], 0, 0, 0);
                    if (p < AH * AW) {  // (a lane beyond the region computed on whatever its addresses held: unused)
                        // outside the image the activation is the second convolution's zero padding: border tiles scale by
                        // 0 or 1 (relu6(...) is finite and >= 0, so x * 1 = x and x * 0 = +0 exactly)
                        float inside = 1.f;
                        if (!INTERIOR) {
                            const int gx = x0 - 1 + rx, gy = y0 - 1 + ry;
                            inside = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? 1.f : 0.f;
                        }
#pragma unroll
                        for (int t = 0; t < NT1; ++t) {
                            half4 o;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float r = __builtin_amdgcn_fmed3f(acc[t][i], 0.f, 6.f);  // ReLU6
                                o[i] = (_Float16)(INTERIOR ? r : r * inside);
                            }
                            *reinterpret_cast<half4*>(s_act + __umul24((unsigned)p, AS) + t * 16 + kg * 4) = o;
                        }
                    }
                    p += 64;
                    rx += 64 - AW;
                    ry += 1;
                    if (rx >= AW) {
                        rx -= AW;
                        ry += 1;
                    }
                }
            }
        };
        if (tile_interior)  // workgroup-uniform
            layer1(std::true_type{});
        else
            layer1(std::false_type{});
    }
#ifdef RTO_NET_DBG_STAMP
    if (x0 == tx_first * kGW) st[3] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef RTO_NET_DBG_STAMP
    if (x0 == tx_first * kGW) st[4] = __builtin_amdgcn_s_memtime();
#endif

    // ---- stage C: layer 2 on the kGH x kGW tile, softmax, stores
    // A wave owns 16 columns x 4 rows of the tile and walks the 6 activation rows under them: each row's three
    // B fragments (kx = 0..2; one ds_read_b128 per lane each) feed the MFMAs of up to three output rows (ky = row -
    // output row), so a group costs 4.5 fragment reads instead of 9 -- the stage was bound by LDS reads (144 KB per
    // tile against 576 clocks of MFMA).  Every accumulator still receives its taps in the order 0..8.
    {
        static_assert(kGW == 32 && kGH == 8, "stage C assigns (column block, row half) = wave");
        const float4v bias = *reinterpret_cast<const float4v*>(s_b2 + kg * 4);
#ifndef RTO_NET_DBG_CREP
#define RTO_NET_DBG_CREP 1
#endif
        for (int rep = 0; rep < RTO_NET_DBG_CREP; ++rep) {
            if (RTO_NET_DBG_CREP > 1) asm volatile("" ::: "memory");
            const int ox = (wave & 1) * 16 + col, oy0 = (wave >> 1) * 4;
            float4v acc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = bias;  // the bias: the first MFMA's C operand
            const _Float16* arow = s_act + __umul24(__umul24((unsigned)oy0, AW) + ox, AS) + kg * 8;
            half8 cur[3], nxt[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) cur[kx] = *reinterpret_cast<const half8*>(arow + kx * AS);
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                if (a < 5) {  // the next activation row is on its way while this one is multiplied
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) nxt[kx] = *reinterpret_cast<const half8*>(arow + ((a + 1) * AW + kx) * AS);
                }
                // (kx outside, output row inside: consecutive MFMAs write different accumulators, and each accumulator
                //  still sees ky = 0..2, kx = 0..2 in order)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ky = a - r;
                        if (ky >= 0 && ky < 3)
                            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ky * 3 + kx], cur[kx], acc[r], 0, 0, 0);
                    }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) cur[kx] = nxt[kx];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gx = x0 + ox, gy = y0 + oy0 + r;
                if (gx < W && gy < H && kg * 4 < 2 * L) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (float)(_Float16)__builtin_amdgcn_fmed3f(acc[r][i], 0.f, 6.f);  // ReLU6 -> fp16 activations, then .float()
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
    }
#ifdef RTO_NET_DBG_STAMP
    if (x0 == tx_first * kGW) {
        st[5] = __builtin_amdgcn_s_memtime();
        if ((blockIdx.x == 3 || blockIdx.x == 11) && (blockIdx.y == 40 || blockIdx.y == 41) && blockIdx.z == 2 && lane == 0)
            printf("blk %d,%d wave %d: fetch+wait+cvt %llu | sync %llu | B %llu | sync %llu | C %llu | start %llu\n", blockIdx.x, blockIdx.y, wave,
                   st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[0]);
    }
#endif
    }  // strip
}

}  // namespace

hipError_t launch_guidance_net(const float* aux, const void* w1, const void* w2, const float* b2, int c1,
                               int levels, int n, int H, int W, float* weight_out, float* guidance_out,
                               int in_mode, const uint32_t* tile_mask, int mask_words, const uint32_t* fill_k,
                               const float* fill_planes, int sparse, float background, hipStream_t stream) {
    if (c1 != 32 || levels != 4 || in_mode < 0 || in_mode > 2) return hipErrorInvalidValue;  // the reference configuration (blender.txt:21-25)
    const int tiles_x = (W + kGW - 1) / kGW;
    // strips of kStrip tiles when that still leaves every CU its workgroups (a batch of frames), shorter ones for a lone frame
    const int tiles_y = (H + kGH - 1) / kGH;
    int strip = kStrip;
    while (strip > 1 && (int64_t)((tiles_x + strip - 1) / strip) * tiles_y * n < 2048) --strip;
    const dim3 grid((tiles_x + strip - 1) / strip, tiles_y, n), block(256);
    const bool pack = guidance_out == nullptr;  // weight_out is then the packed fp16 buffer [n][H][W][8]
    NetCull cull;
    cull.mask = (pack ? fill_k != nullptr : fill_planes != nullptr) ? tile_mask : nullptr;
    cull.mask_words = mask_words;
    cull.tiles_x = (W + 7) / 8;
    cull.fill = cull.mask && pack ? make_uint4(fill_k[0], fill_k[1], fill_k[2], fill_k[3]) : make_uint4(0u, 0u, 0u, 0u);
    for (int i = 0; i < 8; ++i) cull.planes[i] = cull.mask && !pack ? fill_planes[i] : 0.f;
    cull.sparse = sparse && cull.mask && pack && in_mode == 2 ? 1 : 0;
    if (sparse && !cull.sparse) return hipErrorInvalidValue;  // (sparse frames come with tile marks, as the RGBA image, on the packed route)
    cull.bg = background;
#define RTO_NET(IN, PK)                                                                                                   \
    hipLaunchKernelGGL((guidance_fused<32, 4, IN, PK>), grid, block, 0, stream, aux, (const _Float16*)w1, (const _Float16*)w2,     \
                       b2, weight_out, guidance_out, H, W, cull, strip)
    if (pack) {
        if (in_mode == 2) RTO_NET(2, true); else if (in_mode == 1) RTO_NET(1, true); else RTO_NET(0, true);
    } else {
        if (in_mode == 2) RTO_NET(2, false); else if (in_mode == 1) RTO_NET(1, false); else RTO_NET(0, false);
    }
#undef RTO_NET
    return hipGetLastError();
}

}  // namespace rto
