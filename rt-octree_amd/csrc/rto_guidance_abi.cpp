// rto_guidance_abi.cpp -- C ABI of the fused GuidanceNet forward (include/rto.h, guidance_kernels.hip).
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rto.h"
#include "rto_launch.h"
#include "rto_denoise_launch.h"

extern "C" const char* rto_last_error(void);

struct rto_guidance_net {
    int device = 0, c1 = 0, levels = 0;
    void* w1 = nullptr;   // fp16 [c1][96]
    void* w2 = nullptr;   // fp16 [16][9*c1]
    float* b2 = nullptr;  // [16]
    void* packed = nullptr;      // scratch of the packed route: fp16 [n][H][W][8]
    size_t packed_bytes = 0;
    int packed_n = 0, packed_h = 0, packed_w = 0;  // what the scratch currently holds
    // rto_filtering_packed_culled: the filter's output tile over pure background of brightness fill_bg (see ensure_fill_tile)
    std::mutex fill_mu;
    float* fill_tile = nullptr;  // device [32][32][4] (factorised filter) then [8][32][4] (exact filter)
    float fill_planes[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // the background maps as fp32 planes hold them: 4 softmax weights, 4 guidance values
    bool packed_sparse = false;  // the packed maps hold nothing for the tiles the network skipped (RTO_NET_INPUT_SPARSE)
    uint32_t fill_k[4] = {0, 0, 0, 0};  // ... and the network's 8 fp16 outputs for a pixel whose neighbourhood is background
    float fill_bg = 0.f;
    bool fill_valid = false;
    float* planes = nullptr;  // rto_denoise(EXACT): weight + guidance planes, 2 x [n][4][H][W]
    size_t planes_bytes = 0;
};

namespace {
// rto_abi.cpp owns the thread-local error string; this translation unit reports through it
extern "C" int rto_set_error_(int code, const char* msg);
int fail(int code, const std::string& m) { return rto_set_error_(code, m.c_str()); }

// the network's weights live on net->device: launch (and free) there, whatever device is current
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
}  // namespace

extern "C" {

int rto_guidance_net_create(const float* w1, const float* b1, const float* w2, const float* b2, int c1, int levels,
                            int device, rto_guidance_net** out) {
    if (!w1 || !b1 || !w2 || !b2 || !out) return fail(RTO_E_INVALID, "rto_guidance_net_create: null argument");
    if (c1 != 32 || levels != 4)
        return fail(RTO_E_UNSUPPORTED, "fused GuidanceNet supports mid_channels = 32, kernel_levels = 4 (configs/blender.txt)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RTO_E_HIP, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(RTO_E_INVALID, "device index out of range");
    const int cout = 2 * levels;
    // pack: k = tap*Cin + ci, tap = ky*3 + kx; layer 1 rows padded to 96 (12 taps), layer 2 to 16 rows
    std::vector<_Float16> p1((size_t)c1 * 96, (_Float16)0.f), p2((size_t)16 * 9 * c1, (_Float16)0.f);
    for (int co = 0; co < c1; ++co)
        for (int ci = 0; ci < 8; ++ci)
            for (int t = 0; t < 9; ++t) p1[(size_t)co * 96 + t * 8 + ci] = (_Float16)w1[((size_t)co * 8 + ci) * 9 + t];
    // the reference runs the module after `.half()` (network.py:194-201): its biases are fp16 values too.  Layer 1's
    // occupies the first padding slot of its weight row (k = 72; the kernel multiplies it by a constant 1).
    for (int co = 0; co < c1; ++co) p1[(size_t)co * 96 + 72] = (_Float16)b1[co];
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < c1; ++ci)
            for (int t = 0; t < 9; ++t)
                p2[(size_t)co * 9 * c1 + (size_t)t * c1 + ci] = (_Float16)w2[((size_t)co * c1 + ci) * 9 + t];
    std::vector<float> pb2(16, 0.f);
    for (int co = 0; co < cout; ++co) pb2[co] = (float)(_Float16)b2[co];

    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(device);
    auto n = new rto_guidance_net();
    n->device = device;
    n->c1 = c1;
    n->levels = levels;
    bool ok = hipMalloc(&n->w1, p1.size() * 2) == hipSuccess && hipMalloc(&n->w2, p2.size() * 2) == hipSuccess &&
              hipMalloc((void**)&n->b2, 16 * sizeof(float)) == hipSuccess;
    ok = ok && hipMemcpy(n->w1, p1.data(), p1.size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(n->w2, p2.data(), p2.size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(n->b2, pb2.data(), 16 * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    (void)hipSetDevice(prev);
    if (!ok) {
        rto_guidance_net_free(n);
        return fail(RTO_E_HIP, "uploading the GuidanceNet weights failed");
    }
    *out = n;
    return RTO_OK;
}

// flags of the forward calls -> launch_guidance_net's in_mode
static int net_in_mode(int flags) { return (flags & RTO_NET_INPUT_RGBA) ? 2 : (flags & RTO_NET_AUX_SQUARES_IMPLIED) ? 1 : 0; }

int rto_guidance_net_forward(const rto_guidance_net* net, void* stream, const float* aux, int n, int H, int W,
                             float* weight_map, float* guidance_map) {
    return rto_guidance_net_forward_ex(net, stream, aux, n, H, W, weight_map, guidance_map, 0);
}

int rto_guidance_net_forward_ex(const rto_guidance_net* net, void* stream, const float* aux, int n, int H, int W,
                                float* weight_map, float* guidance_map, int flags) {
    if (!net || !aux || !weight_map || !guidance_map || n < 1 || H < 1 || W < 1)
        return fail(RTO_E_INVALID, "rto_guidance_net_forward: bad argument");
    DeviceScope scope(net->device);
    const hipError_t e = rto::launch_guidance_net(aux, net->w1, net->w2, net->b2, net->c1, net->levels, n, H, W,
                                                  weight_map, guidance_map, net_in_mode(flags), nullptr, 0, nullptr, nullptr, 0, 0.f,
                                                  (hipStream_t)stream);
    if (e != hipSuccess) return fail(RTO_E_HIP, std::string("GuidanceNet launch failed: ") + hipGetErrorString(e));
    return RTO_OK;
}

// grows the scratch of the packed route to n x H x W pixels.  Growing frees the old buffer, which earlier work may still
// read: that is the one place this file synchronises -- callers that must stay asynchronous reserve up front.
static int reserve_packed(rto_guidance_net* net, int n, int H, int W) {
    const size_t need = (size_t)n * H * W * 16;
    if (need <= net->packed_bytes) return RTO_OK;
    if (net->packed) {
        if (hipDeviceSynchronize() != hipSuccess || hipFree(net->packed) != hipSuccess) return fail(RTO_E_HIP, "hipFree failed");
        net->packed = nullptr;
        net->packed_bytes = 0;
        net->packed_n = 0;
    }
    if (hipMalloc(&net->packed, need) != hipSuccess) return fail(RTO_E_HIP, "hipMalloc(packed maps) failed");
    net->packed_bytes = need;
    return RTO_OK;
}

static int pointer_device(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return a.device;
}

int rto_guidance_net_reserve(rto_guidance_net* net, int n, int H, int W) {
    if (!net || n < 1 || H < 1 || W < 1) return fail(RTO_E_INVALID, "rto_guidance_net_reserve: bad argument");
    DeviceScope scope(net->device);
    return reserve_packed(net, n, H, W);
}

static int ensure_fill_tile(rto_guidance_net* net, float bg, hipStream_t stream);
static int check_marks(const char* who, const rto_guidance_net* net, const uint32_t* tile_marks, int words_per_frame, int H, int W) {
    const int tiles = ((W + 7) / 8) * ((H + 7) / 8);
    if (words_per_frame != (tiles + 31) / 32 + 1)
        return fail(RTO_E_INVALID, std::string(who) + ": the tile marks are not those of a " + std::to_string(H) + " x " + std::to_string(W) +
                                       " frame (rto_ctx_tile_marks)");
    if (pointer_device(tile_marks) != net->device)
        return fail(RTO_E_INVALID, std::string(who) + ": the tile marks are not memory of the network's device");
    return RTO_OK;
}

int rto_guidance_net_forward_packed_culled(rto_guidance_net* net, void* stream, const float* aux, int n, int H, int W, int flags,
                                           const uint32_t* tile_marks, int words_per_frame, float background) {
    if (!net || !aux || n < 1 || H < 1 || W < 1) return fail(RTO_E_INVALID, "rto_guidance_net_forward_packed: bad argument");
    if (pointer_device(aux) != net->device)
        return fail(RTO_E_INVALID, "rto_guidance_net_forward_packed: aux is not memory of the network's device");
    DeviceScope scope(net->device);
    if (const int rc = reserve_packed(net, n, H, W)) return rc;
    if (tile_marks) {
        if (const int rc = check_marks("rto_guidance_net_forward_packed_culled", net, tile_marks, words_per_frame, H, W)) return rc;
        if (const int rc = ensure_fill_tile(net, background, (hipStream_t)stream)) return rc;
    }
    const bool sparse = (flags & RTO_NET_INPUT_SPARSE) != 0;
    if (sparse && (!tile_marks || !(flags & RTO_NET_INPUT_RGBA)))
        return fail(RTO_E_INVALID, "rto_guidance_net_forward_packed_culled: RTO_NET_INPUT_SPARSE needs the tile marks of the launch and RTO_NET_INPUT_RGBA");
    const hipError_t e = rto::launch_guidance_net(aux, net->w1, net->w2, net->b2, net->c1, net->levels, n, H, W,
                                                  (float*)net->packed, nullptr, net_in_mode(flags),
                                                  tile_marks, words_per_frame, tile_marks ? net->fill_k : nullptr, nullptr, sparse ? 1 : 0, background,
                                                  (hipStream_t)stream);
    if (e != hipSuccess) return fail(RTO_E_HIP, std::string("GuidanceNet launch failed: ") + hipGetErrorString(e));
    net->packed_sparse = sparse;
    net->packed_n = n;
    net->packed_h = H;
    net->packed_w = W;
    return RTO_OK;
}

int rto_guidance_net_forward_packed(rto_guidance_net* net, void* stream, const float* aux, int n, int H, int W, int flags) {
    return rto_guidance_net_forward_packed_culled(net, stream, aux, n, H, W, flags, nullptr, 0, 0.f);
}

int rto_filtering_packed(const rto_guidance_net* net, void* stream, const float* img_in, float* img_out, int n, int H, int W) {
    if (!net || !img_in || !img_out || img_in == img_out) return fail(RTO_E_INVALID, "rto_filtering_packed: bad argument");
    if (!net->packed || net->packed_n < 1) return fail(RTO_E_INVALID, "rto_filtering_packed: no packed maps (call rto_guidance_net_forward_packed first)");
    // the images are the caller's: their extent must be the one the maps were computed for, or the kernel would read and
    // write past a smaller buffer
    if (n != net->packed_n || H != net->packed_h || W != net->packed_w)
        return fail(RTO_E_INVALID, "rto_filtering_packed: " + std::to_string(n) + " x " + std::to_string(H) + " x " + std::to_string(W) +
                                       " images, but the packed maps hold " + std::to_string(net->packed_n) + " x " +
                                       std::to_string(net->packed_h) + " x " + std::to_string(net->packed_w));
    if (pointer_device(img_out) != net->device || pointer_device(img_in) != net->device)
        return fail(RTO_E_INVALID, "rto_filtering_packed: the images are not memory of the network's device");
    if (net->packed_sparse)
        return fail(RTO_E_INVALID, "rto_filtering_packed: the packed maps are sparse (RTO_NET_INPUT_SPARSE): the filter needs the same tile marks "
                                   "(rto_filtering_packed_culled)");
    DeviceScope scope(net->device);
    const hipError_t e = rto::launch_filter_fast_packed(net->packed, net->packed_h, net->packed_w, net->packed_n, img_in, img_out,
                                                        nullptr, 0, nullptr, 0, 0.f, nullptr, (hipStream_t)stream);
    if (e != hipSuccess) return fail(RTO_E_HIP, std::string("filter launch failed: ") + hipGetErrorString(e));
    return RTO_OK;
}

// The output tile of a filter workgroup that sees nothing but background: run the two real kernels once on a synthetic
// 96 x 96 background frame (aux = what write_pixel stores for a ray that meets nothing: colour = bg, alpha = 0) and keep
// the tile of the centre workgroup, whose staged region + the network's receptive field lie inside the frame.  Any other
// such workgroup executes the same instructions on the same values, thread for thread.
static int ensure_fill_tile(rto_guidance_net* net, float bg, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(net->fill_mu);
    if (net->fill_valid && std::memcmp(&net->fill_bg, &bg, sizeof(float)) == 0) return RTO_OK;
    constexpr int S = 3 * rto::kFilterFillSide, T = rto::kFilterFillSide;
    const size_t px = (size_t)S * S;
    std::vector<float> aux(8 * px, 0.f), img(4 * px);
    for (size_t i = 0; i < px; ++i) {
        for (int c = 0; c < 3; ++c) {
            aux[c * px + i] = bg;
            aux[(4 + c) * px + i] = bg * bg;
            img[4 * i + c] = bg;
        }
        img[4 * i + 3] = 1.f;
    }
    float *d_aux = nullptr, *d_img = nullptr, *d_out = nullptr, *d_w = nullptr, *d_g = nullptr;
    void* d_maps = nullptr;
    auto cleanup = [&] {
        for (void* p : {(void*)d_aux, (void*)d_img, (void*)d_out, d_maps, (void*)d_w, (void*)d_g})
            if (p) (void)hipFree(p);
    };
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { return (e = r) == hipSuccess; };
    constexpr int TE = rto::kFilterExactFillH, YE = (S / 2 / TE) * TE;  // the exact filter's tile: 32 x 8; its row in the frame
    if (!net->fill_tile && !ok(hipMalloc((void**)&net->fill_tile, (size_t)(T + TE) * T * 4 * sizeof(float))))
        return fail(RTO_E_HIP, std::string("fill tile: ") + hipGetErrorString(e));
    if (!ok(hipMalloc((void**)&d_aux, aux.size() * sizeof(float))) || !ok(hipMalloc((void**)&d_img, img.size() * sizeof(float))) ||
        !ok(hipMalloc((void**)&d_out, img.size() * sizeof(float))) || !ok(hipMalloc(&d_maps, px * 8 * sizeof(uint16_t))) ||
        !ok(hipMemcpyAsync(d_aux, aux.data(), aux.size() * sizeof(float), hipMemcpyHostToDevice, stream)) ||
        !ok(hipMemcpyAsync(d_img, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice, stream)) ||
        !ok(rto::launch_guidance_net(d_aux, net->w1, net->w2, net->b2, net->c1, net->levels, 1, S, S, (float*)d_maps, nullptr, false, nullptr, 0, nullptr, nullptr, 0, 0.f, stream)) ||
        !ok(rto::launch_filter_fast_packed(d_maps, S, S, 1, d_img, d_out, nullptr, 0, nullptr, 0, 0.f, nullptr, stream)) ||
        !ok(hipMemcpy2DAsync(net->fill_tile, (size_t)T * 4 * sizeof(float), d_out + ((size_t)T * S + T) * 4, (size_t)S * 4 * sizeof(float),
                             (size_t)T * 4 * sizeof(float), T, hipMemcpyDeviceToDevice, stream)) ||
        !ok(hipMemcpyAsync(net->fill_k, (const char*)d_maps + ((size_t)(S / 2) * S + S / 2) * 16, 16, hipMemcpyDeviceToHost, stream)) ||
        // the same through fp32 planes and the exact filter (rto_guidance_net_forward_culled / rto_filtering_culled)
        !ok(hipMalloc((void**)&d_w, 4 * px * sizeof(float))) || !ok(hipMalloc((void**)&d_g, 4 * px * sizeof(float))) ||
        !ok(rto::launch_guidance_net(d_aux, net->w1, net->w2, net->b2, net->c1, net->levels, 1, S, S, d_w, d_g, false, nullptr, 0, nullptr, nullptr, 0, 0.f, stream)) ||
        !ok(rto::launch_filter(d_w, d_g, net->levels, S, S, 1, d_img, d_out, stream)) ||
        !ok(hipMemcpy2DAsync(net->fill_tile + (size_t)T * T * 4, (size_t)T * 4 * sizeof(float), d_out + ((size_t)YE * S + T) * 4,
                             (size_t)S * 4 * sizeof(float), (size_t)T * 4 * sizeof(float), TE, hipMemcpyDeviceToDevice, stream)) ||
        !ok(hipMemcpy2DAsync(net->fill_planes, sizeof(float), d_w + (size_t)(S / 2) * S + S / 2, px * sizeof(float), sizeof(float), 4,
                             hipMemcpyDeviceToHost, stream)) ||
        !ok(hipMemcpy2DAsync(net->fill_planes + 4, sizeof(float), d_g + (size_t)(S / 2) * S + S / 2, px * sizeof(float), sizeof(float), 4,
                             hipMemcpyDeviceToHost, stream)) ||
        !ok(hipStreamSynchronize(stream))) {
        cleanup();
        net->fill_valid = false;
        return fail(RTO_E_HIP, std::string("fill tile: ") + hipGetErrorString(e));
    }
    cleanup();
    net->fill_bg = bg;
    net->fill_valid = true;
    return RTO_OK;
}

int rto_filtering_packed_culled(rto_guidance_net* net, void* stream, const float* img_in, float* img_out, int n, int H, int W,
                                const uint32_t* tile_marks, int words_per_frame, float background) {
    if (!tile_marks) return rto_filtering_packed(net, stream, img_in, img_out, n, H, W);
    if (!net || !img_in || !img_out || img_in == img_out) return fail(RTO_E_INVALID, "rto_filtering_packed_culled: bad argument");
    if (!net->packed || net->packed_n < 1)
        return fail(RTO_E_INVALID, "rto_filtering_packed_culled: no packed maps (call rto_guidance_net_forward_packed first)");
    if (n != net->packed_n || H != net->packed_h || W != net->packed_w)
        return fail(RTO_E_INVALID, "rto_filtering_packed_culled: " + std::to_string(n) + " x " + std::to_string(H) + " x " + std::to_string(W) +
                                       " images, but the packed maps hold " + std::to_string(net->packed_n) + " x " +
                                       std::to_string(net->packed_h) + " x " + std::to_string(net->packed_w));
    if (const int rc = check_marks("rto_filtering_packed_culled", net, tile_marks, words_per_frame, H, W)) return rc;
    if (pointer_device(img_out) != net->device || pointer_device(img_in) != net->device)
        return fail(RTO_E_INVALID, "rto_filtering_packed_culled: the images are not memory of the network's device");
    DeviceScope scope(net->device);
    if (const int rc = ensure_fill_tile(net, background, (hipStream_t)stream)) return rc;
    const hipError_t e = rto::launch_filter_fast_packed(net->packed, net->packed_h, net->packed_w, net->packed_n, img_in, img_out,
                                                        tile_marks, words_per_frame, net->fill_tile, net->packed_sparse ? 1 : 0, background,
                                                        net->fill_k, (hipStream_t)stream);
    if (e != hipSuccess) return fail(RTO_E_HIP, std::string("filter launch failed: ") + hipGetErrorString(e));
    return RTO_OK;
}

int rto_guidance_net_forward_culled(rto_guidance_net* net, void* stream, const float* aux, int n, int H, int W, float* weight_map,
                                    float* guidance_map, int flags, const uint32_t* tile_marks, int words_per_frame, float background) {
    if (!tile_marks) return rto_guidance_net_forward_ex(net, stream, aux, n, H, W, weight_map, guidance_map, flags);
    if (!net || !aux || !weight_map || !guidance_map || n < 1 || H < 1 || W < 1)
        return fail(RTO_E_INVALID, "rto_guidance_net_forward_culled: bad argument");
    if (const int rc = check_marks("rto_guidance_net_forward_culled", net, tile_marks, words_per_frame, H, W)) return rc;
    DeviceScope scope(net->device);
    if (const int rc = ensure_fill_tile(net, background, (hipStream_t)stream)) return rc;
    const hipError_t e = rto::launch_guidance_net(aux, net->w1, net->w2, net->b2, net->c1, net->levels, n, H, W, weight_map, guidance_map,
                                                  net_in_mode(flags), tile_marks, words_per_frame, nullptr,
                                                  net->fill_planes, 0, 0.f, (hipStream_t)stream);
    if (e != hipSuccess) return fail(RTO_E_HIP, std::string("GuidanceNet launch failed: ") + hipGetErrorString(e));
    return RTO_OK;
}

int rto_filtering_culled(rto_guidance_net* net, void* stream, const float* weight_map, const float* guidance_map, int H, int W, int n,
                         const float* img_in, float* img_out, int mode, const uint32_t* tile_marks, int words_per_frame, float background) {
    if (!net) return fail(RTO_E_INVALID, "rto_filtering_culled: null network handle");
    if (!tile_marks) return rto_filtering_batch_mode(stream, weight_map, guidance_map, net->levels, H, W, n, img_in, img_out, mode);
    if (!weight_map || !guidance_map || !img_in || !img_out || img_in == img_out || H < 1 || W < 1 || n < 1)
        return fail(RTO_E_INVALID, "rto_filtering_culled: bad argument");
    if (mode != RTO_FILTER_EXACT && mode != RTO_FILTER_FACTORISED) return fail(RTO_E_INVALID, "rto_filtering_culled: unknown mode");
    if (const int rc = check_marks("rto_filtering_culled", net, tile_marks, words_per_frame, H, W)) return rc;
    if (pointer_device(img_out) != net->device || pointer_device(img_in) != net->device)
        return fail(RTO_E_INVALID, "rto_filtering_culled: the images are not memory of the network's device");
    DeviceScope scope(net->device);
    if (const int rc = ensure_fill_tile(net, background, (hipStream_t)stream)) return rc;
    constexpr size_t kExactTile = (size_t)rto::kFilterFillSide * rto::kFilterFillSide * 4;
    const hipError_t e = mode == RTO_FILTER_EXACT
                             ? rto::launch_filter_culled(weight_map, guidance_map, net->levels, H, W, n, img_in, img_out, tile_marks,
                                                         words_per_frame, net->fill_tile + kExactTile, (hipStream_t)stream)
                             : rto::launch_filter_fast_culled(weight_map, guidance_map, net->levels, H, W, n, img_in, img_out, tile_marks,
                                                              words_per_frame, net->fill_tile, (hipStream_t)stream);
    if (e != hipSuccess) return fail(RTO_E_HIP, std::string("filter launch failed: ") + hipGetErrorString(e));
    return RTO_OK;
}

int rto_denoise(rto_guidance_net* net, rto_ctx* ctx, int n, int mode, void* stream) {
    if (!net || !ctx || n < 1) return fail(RTO_E_INVALID, "rto_denoise: bad argument");
    if (mode != RTO_FILTER_EXACT && mode != RTO_FILTER_FACTORISED) return fail(RTO_E_INVALID, "rto_denoise: unknown mode");
    const int W = rto_ctx_width(ctx), H = rto_ctx_height(ctx), sel = rto_ctx_selected_frame(ctx);
    if (sel + n > rto_ctx_frames(ctx))
        return fail(RTO_E_INVALID, "rto_denoise: " + std::to_string(n) + " frames from slot " + std::to_string(sel) + " of a context of " +
                                       std::to_string(rto_ctx_frames(ctx)));
    const float* aux = rto_ctx_aux(ctx);
    const float* noisy = rto_ctx_noisy(ctx);
    float* image = rto_ctx_image(ctx);
    // the tile marks of the launch that rendered these frames, if it was a batched one over exactly these slots
    const uint32_t* marks = nullptr;
    int words = 0, first = 0, frames = 0;
    float bg = 0.f;
    if (rto_ctx_tile_marks(ctx, &marks, &words, &first, &frames, &bg) != RTO_OK || first != sel || frames < n) marks = nullptr;
    // frames of a lean batched launch (rto_ctx_set_lean_outputs): their aux planes were not written, the noisy image
    // carries r, g, b, alpha -- the network reads that
    int net_flags = RTO_NET_AUX_SQUARES_IMPLIED;
    const int lean = rto_ctx_frames_lean_level(ctx, sel, n);  // 0 full, 1 lean, 2 sparse; -1: the slots disagree
    if (lean < 0)
        return fail(RTO_E_INVALID, "rto_denoise: slots " + std::to_string(sel) + ".." + std::to_string(sel + n - 1) +
                                       " mix full, lean and sparse outputs (a later launch rewrote some of them): denoise each run of slots by itself");
    if (lean) {
        aux = noisy;
        net_flags = RTO_NET_INPUT_RGBA;
        if (lean == 2) {  // sparse: nothing was stored for the pixels of culled tiles
            if (!marks)
                return fail(RTO_E_INVALID, "rto_denoise: sparse lean frames need the tile marks of the launch that rendered them (another launch "
                                           "into this context replaced them)");
            if (mode != RTO_FILTER_FACTORISED)
                return fail(RTO_E_UNSUPPORTED, "rto_denoise: sparse lean frames (rto_ctx_set_lean_outputs level 2) take the factorised route only");
            net_flags |= RTO_NET_INPUT_SPARSE;
        }
    }
    if (mode == RTO_FILTER_FACTORISED) {
        if (const int rc = rto_guidance_net_forward_packed_culled(net, stream, aux, n, H, W, net_flags, marks, words, bg)) return rc;
        return rto_filtering_packed_culled(net, stream, noisy, image, n, H, W, marks, words, bg);
    }
    const size_t plane_floats = (size_t)n * net->levels * H * W;
    {
        DeviceScope scope(net->device);
        if (net->planes_bytes < 2 * plane_floats * sizeof(float)) {
            if (net->planes && (hipDeviceSynchronize() != hipSuccess || hipFree(net->planes) != hipSuccess)) return fail(RTO_E_HIP, "hipFree failed");
            net->planes = nullptr;
            net->planes_bytes = 0;
            if (hipMalloc((void**)&net->planes, 2 * plane_floats * sizeof(float)) != hipSuccess) return fail(RTO_E_HIP, "hipMalloc(map planes) failed");
            net->planes_bytes = 2 * plane_floats * sizeof(float);
        }
    }
    float *wm = net->planes, *gm = net->planes + plane_floats;
    if (const int rc = rto_guidance_net_forward_culled(net, stream, aux, n, H, W, wm, gm, net_flags, marks, words, bg)) return rc;
    return rto_filtering_culled(net, stream, wm, gm, H, W, n, noisy, image, RTO_FILTER_EXACT, marks, words, bg);
}

void rto_guidance_net_free(rto_guidance_net* net) {
    if (!net) return;
    DeviceScope scope(net->device);
    if (net->packed) (void)hipFree(net->packed);
    if (net->fill_tile) (void)hipFree(net->fill_tile);
    if (net->planes) (void)hipFree(net->planes);
    if (net->w1) (void)hipFree(net->w1);
    if (net->w2) (void)hipFree(net->w2);
    if (net->b2) (void)hipFree(net->b2);
    delete net;
}

}  // extern "C"
