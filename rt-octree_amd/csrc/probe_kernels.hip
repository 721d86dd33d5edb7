// probe_kernels.hip -- calibration kernels for the rocprofv3 HBM counters (MI355X_MICROARCH.md "HBM":
// FETCH_SIZE is only calibrated for wide coalesced streams; other access widths must be calibrated on
// a known byte count in the access pattern at hand).  The render path's pattern is one 4-byte load
// per lane from scattered 32-byte nodes, so the probe below issues exactly that: every lane reads
// ONE dword from its own, never repeated 128-byte line of a buffer far larger than the 256 MiB
// Infinity Cache.  Known traffic = lines * (64 or 128) bytes; the counter tells which.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rto.h"

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
