// filter_repro.hip -- second stand-alone check (see pkfma_repro.hip for the first): the library's bit-exact filter kernel
// itself, compiled into this program from its source with packed FMAs (-DRTO_FILTER_PK=1) or scalar ones (=0), on seeded
// pseudo-random inputs -- every launch must return the bits of the first one -- while 7 other processes run pkfma_repro's
// MFMA load.  (Constant inputs, the first version of this file, came out clean as well, but would not show a fault that
// fetches a neighbouring lane's operand.)  Nothing else of the library is involved (no librto.so, no Python, no PyTorch).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math [-fno-slp-vectorize] -DRTO_FILTER_PK=1|0
//         -I include -I rt-octree_amd/csrc tools/repro/filter_repro.hip -o filter_repro
//   ./filter_repro SECONDS
#include "../../rt-octree_amd/csrc/filter_kernels.hip"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// the "trigger" of the library harness (rto_probe_scratch kind 5): dependent MFMAs and a dynamically indexed private array
__global__ void __launch_bounds__(256, 4) own_mfma_kernel(float* out, int iters, int stride) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    float priv[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) priv[i] = (float)(threadIdx.x + i);
    int idx = (threadIdx.x * 7 + stride) % 48;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        acc += priv[idx];
        priv[(idx + 5) % 48] = acc * 0.5f;
        idx = (idx + stride) % 48;
    }
    h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    f4 c = {acc, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[blockIdx.x * 256 + threadIdx.x] = c[0] + c[1];
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? std::atof(argv[1]) : 10.0;
    const bool own_mfma = argc > 2 && std::strcmp(argv[2], "own_mfma") == 0;  // this process launches the trigger itself
    float* dprobe = nullptr;
    if (own_mfma && hipMalloc((void**)&dprobe, (size_t)4096 * 256 * 4)) return 2;
    const int W = 800, H = 800, n = 4, L = 4;
    const size_t px = (size_t)W * H;
    std::vector<float> w((size_t)n * L * px, 0.25f), g((size_t)n * L * px), img((size_t)n * px * 4);
    uint32_t lcg = 12345u;
    auto rnd = [&] { lcg = lcg * 1664525u + 1013904223u; return (float)(lcg >> 8) * (1.0f / 16777216.0f); };
    for (size_t i = 0; i < g.size(); ++i) g[i] = 6.0f * rnd();  // (GuidanceNet ends in ReLU6)
    for (size_t i = 0; i < img.size(); i += 4) {
        img[i] = rnd();
        img[i + 1] = rnd();
        img[i + 2] = rnd();
        img[i + 3] = 1.f;
    }
    float *dw, *dg, *di, *dout;
    if (hipMalloc((void**)&dw, w.size() * 4) || hipMalloc((void**)&dg, g.size() * 4) || hipMalloc((void**)&di, img.size() * 4) ||
        hipMalloc((void**)&dout, img.size() * 4))
        return 2;
    (void)hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(di, img.data(), img.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> out(img.size()), first;
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0, bad_launches = 0, bad_px = 0, by_quarter[4] = {0, 0, 0, 0};
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        if (own_mfma) {
            hipLaunchKernelGGL(own_mfma_kernel, dim3(4096), dim3(256), 0, nullptr, dprobe, 64, 8);
            (void)hipDeviceSynchronize();
        }
        if (rto::launch_filter(dw, dg, L, H, W, n, di, dout, nullptr) != hipSuccess) return 2;
        (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        if (first.empty()) first = out;
        long bad = 0;
        for (int f = 0; f < n; ++f)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const size_t o = (((size_t)f * H + y) * W + x) * 4;
                    if (std::memcmp(&out[o], &first[o], 16) != 0) {
                        ++bad;
                        ++by_quarter[((y & 1) * 32 + (x & 31)) >> 4];  // lane of the pixel in its wave: (row parity, column in the tile)
                    }
                }
        ++launches;
        bad_px += bad;
        bad_launches += bad != 0;
    }
    std::printf("filter_fused (%s FMAs%s): %ld launches, %ld with a pixel that differs from the first launch, %ld such pixels; by lane quarter "
                "0-15 / 16-31 / 32-47 / 48-63: %ld / %ld / %ld / %ld\n",
                RTO_FILTER_PK ? "packed" : "scalar", own_mfma ? ", MFMA + scratch kernel of its own before every launch" : "", launches, bad_launches, bad_px, by_quarter[0], by_quarter[1], by_quarter[2], by_quarter[3]);
    return 0;
}
