// rto_denoise_launch.h -- host-callable launchers of the denoise stage's gfx950 kernels (filter_kernels.hip,
// guidance_kernels.hip).  Apart from rto_launch.h so that the render kernels' code id (bench.py kernel_code_id) changes with
// the render kernels only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rto {

// denoiser/extension/filtering.cu:108-228,440-470: L levels, support = level + 1
// n images per launch: weight/guidance [n][L][H][W], img_in/img_out [n][H][W][4]
hipError_t launch_filter(const float* weight, const float* guidance, int L, int H, int W, int n, const float* img_in,
                         float* img_out, hipStream_t stream);

// the same filter with the exponentials factorised out of the taps (filter_kernels.hip filter_fast): the
// tolerance path, ~1e-6 relative to launch_filter
hipError_t launch_filter_fast(const float* weight, const float* guidance, int L, int H, int W, int n, const float* img_in,
                              float* img_out, hipStream_t stream);

constexpr int kFilterFillSide = 32;
// launch_filter / launch_filter_fast with the tile skipping described below (launch_filter_fast_packed); the fill tiles are
// [kFilterExactFillH][kFilterFillSide][4] and [kFilterFillSide][kFilterFillSide][4] floats
constexpr int kFilterExactFillH = 8;
hipError_t launch_filter_culled(const float* weight, const float* guidance, int L, int H, int W, int n, const float* img_in,
                                float* img_out, const uint32_t* tile_mask, int mask_words, const float* fill_tile, hipStream_t stream);
hipError_t launch_filter_fast_culled(const float* weight, const float* guidance, int L, int H, int W, int n, const float* img_in,
                                     float* img_out, const uint32_t* tile_mask, int mask_words, const float* fill_tile,
                                     hipStream_t stream);

// ... on the GuidanceNet kernel's packed fp16 maps [n][H][W][8] (4 logits + 4 guidance values), L = 4.
// tile_mask != nullptr: the render context's tile marks of these n frames (FrameBatch::tile_mask); a workgroup whose inputs
// all lie in unmarked (culled = background) tiles copies fill_tile ([32][32][4] floats) instead of filtering
// sparse (round 6): img_in holds no pixel of an unmarked render tile (read as `background`) and packed_maps none of a network
// tile that saw only such tiles (read as fill_maps, the network's 8 fp16 outputs over background)
hipError_t launch_filter_fast_packed(const void* packed_maps, int H, int W, int n, const float* img_in, float* img_out,
                                     const uint32_t* tile_mask, int mask_words, const float* fill_tile, int sparse, float background,
                                     const uint32_t* fill_maps, hipStream_t stream);

// training side: forward that also saves rgb_filtered [n][L][H][W][4], max_map / inv_kernel_sum
// [n][L][H][W] (filtering.cu:205-216), and the backward (filtering.cu:230-301) in gather form
hipError_t launch_filter_train(const float* weight, const float* guidance, int L, int H, int W, int n,
                               const float* img_in, float* img_out, float* rgb_filtered, float* max_map,
                               float* inv_kernel_sum, hipStream_t stream);
hipError_t launch_filter_backward(const float* grad_out, const float* img_in, const float* weight, const float* guidance,
                                  const float* rgb_filtered, const float* max_map, const float* inv_kernel_sum, int L,
                                  int H, int W, int n, float* grad_weight, float* grad_guidance, hipStream_t stream);

// fused compact GuidanceNet (guidance_kernels.hip): w1 fp16 [c1][96], w2 fp16 [16][9*c1], b2 [16];
// guidance_out == nullptr: weight_out receives the packed fp16 maps [n][H][W][8] instead
// sparse (round 6; in_mode 2, packed, tile_mask given): input pixels of unmarked render tiles are taken as (background x 3, 0)
// instead of read, and the skipped tiles' maps are not stored
// in_mode: 0 = aux [n][8][H][W]; 1 = the same, planes 4..7 implied (squares of planes 0..3); 2 = aux is an interleaved image
// [n][H][W][4] = r, g, b, alpha (planes 0..3 of the aux buffer, as a lean batched launch leaves them), squares implied
hipError_t launch_guidance_net(const float* aux, const void* w1, const void* w2, const float* b2, int c1,
                               int levels, int n, int H, int W, float* weight_out, float* guidance_out,
                               int in_mode, const uint32_t* tile_mask, int mask_words, const uint32_t* fill_k,
                               const float* fill_planes, int sparse, float background, hipStream_t stream);

}  // namespace rto
