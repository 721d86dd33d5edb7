// pkfma_repro.hip -- stand-alone check of the observation in DESIGN.md "Determinism when the GPU is shared": does
// v_pk_fma_f32 return wrong values while ANOTHER process keeps the GPU busy with MFMA work?  Nothing of librto is used.
//   hipcc --offload-arch=gfx950 -O2 pkfma_repro.hip -o pkfma_repro
//   ./pkfma_repro load SECONDS          one "other process": dependent MFMA chains on every CU (optionally with a private array)
//   ./pkfma_repro check SECONDS [pk|scalar]   the victim: a kernel of packed (or scalar) FMAs whose result is known exactly;
//                                       prints launches, launches with a wrong lane, and which lanes
// tools/repro/run.sh starts 7 loads and 1 check of each kind.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) mfma_load(float* out, int iters, int stride) {
    float priv[48];  // dynamically indexed: lives in scratch
#pragma unroll
    for (int i = 0; i < 48; ++i) priv[i] = (float)(threadIdx.x + i);
    int idx = (threadIdx.x * 7 + stride) % 48;
    h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    f4 c = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
        c[0] += priv[idx];
        priv[(idx + 5) % 48] = c[1] * 0.5f;
        idx = (idx + stride) % 48;
    }
    out[blockIdx.x * 256 + threadIdx.x] = c[0] + c[1] + c[2] + c[3];
}

// acc_{k+1} = acc_k * m + d per component with m, d and the start the SAME for every lane: every lane must end on the same
// bits.  PK: both components in one v_pk_fma_f32 per step; else two v_fma_f32.
template <bool PK>
__global__ void __launch_bounds__(256) fma_check(float2v* out, int iters, float m, float d) {
    float2v acc = {1.0f, 0.5f};
    const float2v mm = {m, m}, dd = {d, d * 0.75f};
    for (int it = 0; it < iters; ++it) {
        if (PK) {
            acc = __builtin_elementwise_fma(acc, mm, dd);
        } else {
            acc.x = __builtin_fmaf(acc.x, mm.x, dd.x);
            asm volatile("" : "+v"(acc.x));  // (keeps the two FMAs from being paired into one packed instruction)
            acc.y = __builtin_fmaf(acc.y, mm.y, dd.y);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
            return 2;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: pkfma_repro load|check SECONDS [pk|scalar]\n");
        return 2;
    }
    const bool load = std::strcmp(argv[1], "load") == 0;
    const double seconds = std::atof(argv[2]);
    const bool pk = argc < 4 || std::strcmp(argv[3], "pk") == 0;
    const int blocks = 4096;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (load) {
        float* out = nullptr;
        CK(hipMalloc((void**)&out, (size_t)blocks * 256 * sizeof(float)));
        long launches = 0;
        while (elapsed() < seconds) {
            hipLaunchKernelGGL(mfma_load, dim3(blocks), dim3(256), 0, nullptr, out, 256, 3 + (int)(launches % 5));
            CK(hipDeviceSynchronize());
            ++launches;
        }
        std::printf("load: %ld launches\n", launches);
        return 0;
    }
    float2v* out = nullptr;
    CK(hipMalloc((void**)&out, (size_t)blocks * 256 * sizeof(float2v)));
    std::vector<float2v> host((size_t)blocks * 256);
    long launches = 0, bad_launches = 0, bad_values = 0;
    long lane_hist[4] = {0, 0, 0, 0};  // wrong values by lane quarter (0..15, 16..31, 32..47, 48..63)
    while (elapsed() < seconds) {
        if (pk)
            hipLaunchKernelGGL(fma_check<true>, dim3(blocks), dim3(256), 0, nullptr, out, 4096, 0.999f, 0.37f);
        else
            hipLaunchKernelGGL(fma_check<false>, dim3(blocks), dim3(256), 0, nullptr, out, 4096, 0.999f, 0.37f);
        CK(hipMemcpy(host.data(), out, host.size() * sizeof(float2v), hipMemcpyDeviceToHost));
        // the reference: lane 0 of workgroup 0 when most lanes agree with it, else the majority is not worth finding -- report
        const float2v ref = host[0];
        long bad = 0;
        for (size_t i = 0; i < host.size(); ++i)
            if (std::memcmp(&host[i], &ref, sizeof(ref)) != 0) {
                ++bad;
                ++lane_hist[(i & 63) >> 4];
            }
        ++launches;
        bad_values += bad;
        bad_launches += bad != 0;
    }
    std::printf("check %s: %ld launches, %ld with a lane that differs from lane 0, %ld such values; by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %ld / %ld / %ld / %ld\n",
                pk ? "v_pk_fma_f32" : "v_fma_f32", launches, bad_launches, bad_values, lane_hist[0], lane_hist[1], lane_hist[2], lane_hist[3]);
    return 0;
}
