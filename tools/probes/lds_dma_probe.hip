// Standalone probe (development tool, not part of librto): what does global_load_lds_dword write where?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_dma_probe.hip -o rt-octree_amd/bin/lds_dma_probe ; run on an MI355X
// Each of the 4 waves of a workgroup points M0 at a row base and lets its ODD lanes load src[1000 * wave + lane] with an
// instruction offset of 0 or 4 bytes; afterwards every thread dumps the LDS dwords around its column.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef const __attribute__((address_space(1))) unsigned* gp_t;
typedef __attribute__((address_space(3))) unsigned* lp_t;

template <int OFF, bool WAVE_BASE>
__global__ void probe(const unsigned* __restrict__ src, unsigned* __restrict__ out) {
    extern __shared__ unsigned s[];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (int i = tid; i < 4 * 256; i += 256) s[i] = 0xdead0000u + i;
    __syncthreads();
    // row 1 (dwords 256..511); WAVE_BASE: M0 = row + the wave's 64-dword slice, else M0 = row for every wave
    lp_t row = (lp_t)(s + 256 + (WAVE_BASE ? (int)__builtin_amdgcn_readfirstlane((int)(tid & 0xc0u)) : 0));
    if (lane & 1u) __builtin_amdgcn_global_load_lds((gp_t)(src + 1000u * wave + lane), row, 4, OFF, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 4 * 256; i += 256) out[i] = s[i];
}

int main() {
    const int N = 8192;
    std::vector<unsigned> h(N);
    for (int i = 0; i < N; ++i) h[i] = 0x51000000u + i;
    unsigned *src, *out;
    hipMalloc(&src, N * 4);
    hipMalloc(&out, 1024 * 4);
    hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> r(1024);
    for (int variant = 0; variant < 4; ++variant) {
        hipMemset(out, 0, 1024 * 4);
        if (variant == 0) hipLaunchKernelGGL((probe<0, true>), dim3(1), dim3(256), 4096, 0, src, out);
        if (variant == 1) hipLaunchKernelGGL((probe<4, true>), dim3(1), dim3(256), 4096, 0, src, out);
        if (variant == 2) hipLaunchKernelGGL((probe<0, false>), dim3(1), dim3(256), 4096, 0, src, out);
        if (variant == 3) hipLaunchKernelGGL((probe<4, false>), dim3(1), dim3(256), 4096, 0, src, out);
        hipError_t e = hipDeviceSynchronize();
        printf("variant %d (offset %d, %s): %s\n", variant, (variant & 1) * 4, variant < 2 ? "M0 = row + wave slice" : "M0 = row", hipGetErrorString(e));
        hipMemcpy(r.data(), out, 1024 * 4, hipMemcpyDeviceToHost);
        int shown = 0;
        for (int i = 0; i < 1024; ++i)
            if (r[i] != 0xdead0000u + i) {
                if (shown < 12 || (i % 64) < 4) printf("  lds[%4d] (row %d, col %3d) = %08x  (src index %u)\n", i, i / 256, i % 256, r[i], r[i] - 0x51000000u);
                ++shown;
            }
        printf("  %d dwords changed\n", shown);
    }
    return 0;
}
