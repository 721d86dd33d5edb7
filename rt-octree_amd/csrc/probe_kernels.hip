// probe_kernels.hip -- calibration kernels for the rocprofv3 HBM counters (MI355X_MICROARCH.md "HBM":
// FETCH_SIZE is only calibrated for wide coalesced streams; other access widths must be calibrated on
// a known byte count in the access pattern at hand).  The render path's pattern is one 4-byte load
// per lane from scattered 32-byte nodes, so the probe below issues exactly that: every lane reads
// ONE dword from its own, never repeated 128-byte line of a buffer far larger than the 256 MiB
// Infinity Cache.  Known traffic = lines * (64 or 128) bytes; the counter tells which.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rto.h"
#include "rto_device_math.h"

namespace {
__global__ void gather_probe_kernel(const uint32_t* __restrict__ buf, uint64_t n_lines, uint64_t stride_lines,
                                    uint32_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lines) return;
    // a permutation of the lines (stride coprime with n_lines) so neighbouring lanes hit far-apart lines
    const uint64_t line = (i * stride_lines) % n_lines;
    const uint32_t v = buf[line * 32];
    if (v == 0x12345678u) out[0] = v;  // keep the load alive; never true for a zero-filled buffer
}
}  // namespace

extern "C" int rto_probe_gather(uint64_t n_lines, int repeats) {
    uint32_t* buf = nullptr;
    uint32_t* out = nullptr;
    if (hipMalloc((void**)&buf, n_lines * 128) != hipSuccess || hipMalloc((void**)&out, 4) != hipSuccess) return RTO_E_HIP;
    (void)hipMemset(buf, 0, n_lines * 128);
    (void)hipDeviceSynchronize();
    const uint64_t stride = 1000003ULL;  // prime, far larger than a DRAM page
    for (int r = 0; r < repeats; ++r)
        hipLaunchKernelGGL(gather_probe_kernel, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, nullptr, buf,
                           n_lines, stride, out);
    const hipError_t e = hipDeviceSynchronize();
    (void)hipFree(buf);
    (void)hipFree(out);
    return e == hipSuccess ? RTO_OK : RTO_E_HIP;
}

// ---------------------------------------------------------------------------------------------
// Ceiling probes for the traversal kernel (DESIGN.md "What bounds the traversal").
//
// rto_probe_gather_sweep: what rate of scattered dword gathers can a CU's vector L1 (TCP) sustain?
// Every wave-level load touches exactly K distinct 64-byte lines of a `table_bytes` table (lines drawn
// pseudo-randomly per wave and iteration), all 64 lanes active, the lanes dealt over the K lines either
// interleaved (lane % K: no two neighbouring lanes share a line -- the quad-level merge of the L1 finds
// nothing) or blocked (lane * K / 64: neighbouring lanes share a line).  DEP = 1 makes each load's
// address depend on the previous load's value (one gather in flight per wave, the traversal's shape);
// DEP = 0 keeps four independent gathers in flight per wave.  The grid is persistent: `wps` waves per
// SIMD on every CU.  Reported: wall time (HIP events) and the mean shader-clock cycles per wave
// (s_memtime), from which line accesses per clock per CU follow.
namespace {

__device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

template <bool DEP>
__global__ void __launch_bounds__(256) gather_sweep_kernel(const uint32_t* __restrict__ table, uint32_t line_mask,
                                                            int K, int blocked, int iters, uint32_t* __restrict__ sink,
                                                            unsigned long long* __restrict__ cycles) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t j = blocked ? (lane * (uint32_t)K) >> 6 : lane % (uint32_t)K;  // which of the K lines
    const uint32_t dw = (lane * 7u + 3u) & 15u;                                      // dword inside the line
    const uint32_t seed = mix32(gwave * 0x9e3779b9U + 12345u);
    uint32_t acc = 0, dep = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (DEP) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            const uint32_t line = mix32((seed + (uint32_t)it * 0x85ebca6bU) ^ (j * 0xc2b2ae35U)) + dep;
            const uint32_t v = table[(uint64_t)(line & line_mask) * 16u + dw];
            dep = v;  // table is zero-filled: the value is 0, the dependence is real
            acc += v;
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < iters; it += 4) {
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t line = mix32((seed + (uint32_t)(it + u) * 0x85ebca6bU) ^ (j * 0xc2b2ae35U));
                v[u] = table[(uint64_t)(line & line_mask) * 16u + dw];
            }
            acc += v[0] + v[1] + v[2] + v[3];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x12345678u) sink[0] = acc;
    if (lane == 0) atomicAdd(cycles, t1 - t0);
}

// VALU issue-rate probe: 16 independent chains per lane, one op each per body, `kind` picks the op mix:
// 0 = v_fma_f32, 1 = integer (xor / add / bfe), 2 = the traversal's mix (mul, add, med3, cvt, fract, max, min)
__global__ void __launch_bounds__(256) valu_probe_kernel(int kind, int iters, float k0, float k1, float* __restrict__ sink,
                                                          unsigned long long* __restrict__ cycles) {
    float a[16];
    uint32_t u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = k0 * (float)(threadIdx.x + i);
        u[i] = threadIdx.x * 2654435761u + i;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (kind == 0) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], k1, k0);
        }
    } else if (kind == 1) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                u[i] = (u[i] ^ u[i + 1]) + 0x9e3779b9u;
                u[i + 1] = __builtin_amdgcn_ubfe(u[i + 1], 3u, 29u) + u[i];
            }
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                a[i] = __builtin_amdgcn_fmed3f(a[i] * k1 + a[i + 1], 0.f, 0.999999f);
                u[i] = (uint32_t)(a[i] * 16777216.f);
                a[i + 1] = __builtin_amdgcn_fractf(a[i + 1] * k1);
                a[i + 2] = __builtin_fmaxf(a[i + 2] + k0, a[i + 3]);
                a[i + 3] = __builtin_fminf(a[i + 3] * k1, a[i + 2]);
                u[i + 1] = (uint32_t)__clz((int)(u[i] ^ u[i + 1])) + u[i + 1];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        s += a[i];
        x ^= u[i];
    }
    if (s == 1234.5f && x == 77u) sink[0] = s;
    if ((threadIdx.x & 63u) == 0) atomicAdd(cycles, t1 - t0);
}

}  // namespace

// out[0] = wall ms per launch, out[1] = mean shader cycles per wave, out[2] = waves, out[3] = wave-level loads per wave
extern "C" int rto_probe_gather_sweep(uint64_t table_bytes, int lines_per_gather, int blocked, int dependent, int wps,
                                      int iters, int repeats, double out[4]) {
    if (!out || lines_per_gather < 1 || lines_per_gather > 64 || wps < 1 || wps > 8 || iters < 4 || repeats < 1)
        return RTO_E_INVALID;
    uint64_t lines = 1;
    while (lines * 2 * 64 <= table_bytes) lines *= 2;  // power of two lines of 64 B
    uint32_t *table = nullptr, *sink = nullptr;
    unsigned long long* cyc = nullptr;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return RTO_E_HIP;
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    if (hipMalloc((void**)&table, lines * 64) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess ||
        hipMalloc((void**)&cyc, 8) != hipSuccess)
        return RTO_E_HIP;
    (void)hipMemset(table, 0, lines * 64);
    (void)hipMemset(cyc, 0, 8);
    const int blocks = prop.multiProcessorCount * wps;  // 256 threads = 4 waves = one per SIMD
    iters &= ~3;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto launch = [&]() {
        if (dependent)
            hipLaunchKernelGGL(gather_sweep_kernel<true>, dim3(blocks), dim3(256), 0, nullptr, table, (uint32_t)(lines - 1),
                               lines_per_gather, blocked, iters, sink, cyc);
        else
            hipLaunchKernelGGL(gather_sweep_kernel<false>, dim3(blocks), dim3(256), 0, nullptr, table, (uint32_t)(lines - 1),
                               lines_per_gather, blocked, iters, sink, cyc);
    };
    launch();  // warm the caches the table fits in
    (void)hipDeviceSynchronize();
    (void)hipMemset(cyc, 0, 8);
    (void)hipEventRecord(e0, nullptr);
    for (int r = 0; r < repeats; ++r) launch();
    (void)hipEventRecord(e1, nullptr);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long total = 0;
    (void)hipMemcpy(&total, cyc, 8, hipMemcpyDeviceToHost);
    const double waves = (double)blocks * 4.0;
    out[0] = ms / repeats;
    out[1] = (double)total / (waves * repeats);
    out[2] = waves;
    out[3] = iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(table);
    (void)hipFree(sink);
    (void)hipFree(cyc);
    return e == hipSuccess ? RTO_OK : RTO_E_HIP;
}

// -det_log_one_minus(k / 2^23) for k = first_k .. first_k + count - 1: every value a threshold draw can take
// (rt_core.cuh:67-88: t = -logf(1 - rng.next_float())), for the exhaustive comparison with the oracle
__global__ void __launch_bounds__(256) threshold_probe_kernel(uint32_t first_k, uint32_t count, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const float u01 = __uint_as_float(((first_k + i) & 0x7fffffu) | 0x3f800000u) - 1.0f;  // pcg_next_float's mapping
    out[i] = -rto::det_log_one_minus(u01);
}

extern "C" int rto_probe_thresholds(uint32_t first_k, uint32_t count, float* host_out) {
    if (!host_out || count == 0 || (uint64_t)first_k + count > (1ull << 23)) return RTO_E_INVALID;
    float* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)count * 4) != hipSuccess) return RTO_E_HIP;
    hipLaunchKernelGGL(threshold_probe_kernel, dim3((count + 255u) / 256u), dim3(256), 0, nullptr, first_k, count, d);
    const hipError_t e = hipMemcpy(host_out, d, (size_t)count * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return e == hipSuccess ? RTO_OK : RTO_E_HIP;
}

// the device's deterministic math, evaluated on the floats with bit patterns first_bits + i * stride
__global__ void __launch_bounds__(256) math_probe_kernel(int fn, uint32_t first_bits, uint32_t stride, uint32_t count,
                                                         float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const float x = __uint_as_float(first_bits + i * stride);
    out[i] = fn == 0 ? rto::det_logf(x) : fn == 1 ? rto::det_expf(x) : rto::fexp_f32(x);
}

extern "C" int rto_probe_math(int fn, uint32_t first_bits, uint32_t stride, uint32_t count, float* host_out) {
    if (!host_out || count == 0 || fn < 0 || fn > 2) return RTO_E_INVALID;
    float* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)count * 4) != hipSuccess) return RTO_E_HIP;
    hipLaunchKernelGGL(math_probe_kernel, dim3((count + 255u) / 256u), dim3(256), 0, nullptr, fn, first_bits, stride, count, d);
    const hipError_t e = hipMemcpy(host_out, d, (size_t)count * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return e == hipSuccess ? RTO_OK : RTO_E_HIP;
}

// out[0] = wall ms, out[1] = mean shader cycles per wave, out[2] = waves, out[3] = VALU instructions per wave (nominal)
extern "C" int rto_probe_valu(int kind, int wps, int iters, double out[4]) {
    if (!out || wps < 1 || wps > 8 || iters < 1 || kind < 0 || kind > 2) return RTO_E_INVALID;
    int dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return RTO_E_HIP;
    float* sink = nullptr;
    unsigned long long* cyc = nullptr;
    if (hipMalloc((void**)&sink, 4) != hipSuccess || hipMalloc((void**)&cyc, 8) != hipSuccess) return RTO_E_HIP;
    (void)hipMemset(cyc, 0, 8);
    const int blocks = prop.multiProcessorCount * wps;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(valu_probe_kernel, dim3(blocks), dim3(256), 0, nullptr, kind, 16, 1.0001f, 0.9999f, sink, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemset(cyc, 0, 8);
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(valu_probe_kernel, dim3(blocks), dim3(256), 0, nullptr, kind, iters, 1.0001f, 0.9999f, sink, cyc);
    (void)hipEventRecord(e1, nullptr);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long total = 0;
    (void)hipMemcpy(&total, cyc, 8, hipMemcpyDeviceToHost);
    const double waves = (double)blocks * 4.0;
    // per body: kind 0: 16 fma; kind 1: 8 x (xor, add, bfe, add) = 32; kind 2: 4 x (mul, add, med3, mul, cvt, mul, fract,
    // add, max, mul, min, xor, ffbh, add) = 56
    const double per_body = kind == 0 ? 16.0 : kind == 1 ? 32.0 : 56.0;
    out[0] = ms;
    out[1] = (double)total / waves;
    out[2] = waves;
    out[3] = per_body * iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    (void)hipFree(cyc);
    return e == hipSuccess ? RTO_OK : RTO_E_HIP;
}
