// Known-answer generator for pcg32, compiled against the REFERENCE's own header where it lies
// (/root/reference/renderer/3rdparty/pcg32.h; the two CUDA qualifiers are erased with
// -D__host__= -D__device__= on the command line, nothing else is substituted).
// Output: JSON on stdout -> tests/golden/pcg32_kat.json (see Makefile).  Authoring container only.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "pcg32.h"

static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
    printf("{\n");
    // RenderContext::rng = pcg32(20230418)  (render_context.hpp:16)
    pcg32 r(20230418);
    printf("  \"seed\": 20230418,\n  \"state0\": \"%016llx\",\n  \"inc\": \"%016llx\",\n",
           (unsigned long long)r.state, (unsigned long long)r.inc);
    printf("  \"next_uint\": [");
    { pcg32 q = r; for (int i = 0; i < 16; ++i) printf("%s%u", i ? ", " : "", q.next_uint()); }
    printf("],\n  \"next_float_bits\": [");
    { pcg32 q = r; for (int i = 0; i < 16; ++i) printf("%s%u", i ? ", " : "", fbits(q.next_float())); }
    printf("],\n  \"advance\": [\n");
    const long long deltas[] = {0, 1, 2, 5, 6, 63, 64, 4799, 74070, 3839994, 66355199,
                                (1ll << 31) - 1, 1ll << 32, 100ll << 32, 299ll << 32, -1, -74070};
    const int nd = sizeof(deltas) / sizeof(deltas[0]);
    for (int i = 0; i < nd; ++i) {
        pcg32 q = r;
        q.advance(deltas[i]);
        unsigned long long st = q.state;
        uint32_t nu = q.next_uint();
        printf("    {\"delta\": %lld, \"state\": \"%016llx\", \"next_uint\": %u}%s\n", deltas[i], st, nu,
               i + 1 < nd ? "," : "");
    }
    printf("  ],\n  \"frame_pixel\": [\n");
    // per-frame jump (main_headless.cpp:478,506) then per-pixel jump (volrend.cu:157)
    const int frames[] = {0, 1, 100, 299};
    const int pix[] = {0, 1, 12345, 639999};
    const int spps[] = {1, 6, 32};
    int first = 1;
    for (int f : frames) for (int p : pix) for (int s : spps) {
        pcg32 q = r;
        for (int k = 0; k < f; ++k) q.advance();
        q.advance(p * s);
        unsigned long long st = q.state;
        float fl = q.next_float();
        printf("%s    {\"frame\": %d, \"idx\": %d, \"spp\": %d, \"state\": \"%016llx\", \"next_float_bits\": %u}",
               first ? "" : ",\n", f, p, s, st, fbits(fl));
        first = 0;
    }
    printf("\n  ]\n}\n");
    return 0;
}
