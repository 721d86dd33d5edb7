// Reads an .npz with the REFERENCE's vendored cnpy (renderer/3rdparty/cnpy/cnpy.cpp, compiled where
// it lies) and prints, per array: name, word_size, fortran_order, shape, FNV-1a-64 of the raw bytes.
// tests compare our own npz reader against this listing.  Authoring container only.
#include <cstdint>
#include <cstdio>
#include <string>
#include "cnpy.h"

static uint64_t fnv1a(const unsigned char* p, size_t n) {
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ULL; }
    return h;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: cnpy_dump file.npz\n"); return 2; }
    cnpy::npz_t npz = cnpy::npz_load(argv[1]);
    printf("{");
    bool first = true;
    for (auto& kv : npz) {
        const cnpy::NpyArray& a = kv.second;
        printf("%s\n  \"%s\": {\"word_size\": %zu, \"fortran_order\": %d, \"shape\": [", first ? "" : ",",
               kv.first.c_str(), a.word_size, (int)a.fortran_order);
        for (size_t i = 0; i < a.shape.size(); ++i) printf("%s%zu", i ? ", " : "", a.shape[i]);
        printf("], \"nbytes\": %zu, \"fnv1a64\": \"%016llx\"}", a.num_bytes(),
               (unsigned long long)fnv1a((const unsigned char*)a.data<char>(), a.num_bytes()));
        first = false;
    }
    printf("\n}\n");
    return 0;
}
