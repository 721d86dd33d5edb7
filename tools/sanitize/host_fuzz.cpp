// host_fuzz.cpp -- the host-side loaders under AddressSanitizer + UBSan (CPU only; GPU ASan is not available on this pool).
// Build + run: tools/sanitize/run.sh.  Input: a valid tree.npz, a transforms.json, a _poses_bounds.npy (written by run.sh through
// rt_octree_amd.synth).  For each: load it, then load `iters` mutations of it (truncations, byte flips in the header /
// central directory / payload, zeroed ranges, grown length fields).  Every mutation must either load or throw
// std::runtime_error -- anything else (a sanitizer report, a signal, another exception type) fails the run.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../rt-octree_amd/csrc/cli/imwrite.h"
#include "../../rt-octree_amd/csrc/cli/poses.h"
#include "../../rt-octree_amd/csrc/host/n3tree_host.h"
#include "../../rt-octree_amd/csrc/host/npz.h"

namespace {

std::vector<uint8_t> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) {
        std::fprintf(stderr, "cannot read %s\n", p.c_str());
        std::exit(2);
    }
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

void spit(const std::string& p, const std::vector<uint8_t>& b) {
    std::ofstream f(p, std::ios::binary | std::ios::trunc);
    f.write(reinterpret_cast<const char*>(b.data()), (std::streamsize)b.size());
}

std::vector<uint8_t> mutate(const std::vector<uint8_t>& src, std::mt19937_64& rng) {
    std::vector<uint8_t> b = src;
    if (b.empty()) return b;
    auto pick = [&](size_t lo, size_t hi) { return lo + (size_t)(rng() % (hi - lo + 1)); };
    switch (rng() % 6) {
        case 0:  // truncate anywhere (favouring the tail: the zip central directory lives there)
            b.resize(rng() % 2 ? pick(0, b.size()) : b.size() - pick(0, std::min<size_t>(b.size(), 200)));
            break;
        case 1: {  // flip bytes near the end (central directory / end record)
            const size_t span = std::min<size_t>(b.size(), 400);
            for (int k = 0; k < 1 + (int)(rng() % 4); ++k) b[b.size() - 1 - pick(0, span - 1)] ^= (uint8_t)(1u << (rng() % 8));
        } break;
        case 2: {  // flip bytes near the start (first local header + npy header)
            const size_t span = std::min<size_t>(b.size(), 400);
            for (int k = 0; k < 1 + (int)(rng() % 4); ++k) b[pick(0, span - 1)] ^= (uint8_t)(1u << (rng() % 8));
        } break;
        case 3: {  // flip bytes anywhere
            for (int k = 0; k < 1 + (int)(rng() % 8); ++k) b[pick(0, b.size() - 1)] = (uint8_t)rng();
        } break;
        case 4: {  // zero a range
            const size_t a = pick(0, b.size() - 1), n = std::min<size_t>(b.size() - a, pick(1, 4096));
            std::memset(b.data() + a, 0, n);
        } break;
        default: {  // 0xff a short range (length fields become huge)
            const size_t a = pick(0, b.size() - 1), n = std::min<size_t>(b.size() - a, pick(1, 8));
            std::memset(b.data() + a, 0xff, n);
        } break;
    }
    return b;
}

template <typename F>
int fuzz(const char* what, const std::string& path, const std::string& tmp, int iters, uint64_t seed, F load) {
    const std::vector<uint8_t> src = slurp(path);
    load(path);  // the valid file must load (an exception here ends the run)
    std::mt19937_64 rng(seed);
    int ok = 0, refused = 0;
    for (int i = 0; i < iters; ++i) {
        spit(tmp, mutate(src, rng));
        try {
            load(tmp);
            ++ok;
        } catch (const std::runtime_error&) {
            ++refused;
        } catch (const std::exception& e) {  // bad_alloc on a grown length field, out_of_range, ...: not the documented error
            std::fprintf(stderr, "%s: mutation %d raised %s, not std::runtime_error\n", what, i, e.what());
            return 1;
        }
    }
    std::printf("%s: %d mutations: %d loaded, %d refused with std::runtime_error\n", what, iters, ok, refused);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: host_fuzz tree.npz transforms.json poses_bounds.npy tmpdir iters\n");
        return 2;
    }
    const std::string tree = argv[1], json = argv[2], npy = argv[3], tmp = argv[4];
    const int iters = std::atoi(argv[5]);
    int rc = 0;
    rc |= fuzz("tree.npz", tree, tmp + "/m.npz", iters, 1, [](const std::string& p) {
        rto::HostTree t;
        if (t.open(p)) {
            (void)rto::tree_max_depth(t.child, t.capacity, t.N);  // walks every child offset (refuses one out of range)
            volatile uint16_t sink = t.data ? t.data[(size_t)t.capacity * t.N * t.N * t.N * t.data_dim - 1] : 0;  // the last element is mapped
            (void)sink;
        }
    });
    rc |= fuzz("transforms.json", json, tmp + "/transforms_m.json", iters, 2, [](const std::string& p) {
        rto::PoseSet ps;
        rto::load_poses("blender", p, false, ps);
    });
    rc |= fuzz("poses_bounds.npy", npy, tmp + "/poses_bounds_m.npy", iters, 3, [](const std::string& p) {
        rto::PoseSet ps;
        rto::load_poses("llff", p, false, ps);
    });
    {  // a hostile nesting depth must be refused, not recursed into
        spit(tmp + "/transforms_deep.json", std::vector<uint8_t>(2000000, (uint8_t)'['));
        try {
            rto::PoseSet ps;
            rto::load_poses("blender", tmp + "/transforms_deep.json", false, ps);
            rc |= 1;
        } catch (const std::runtime_error& e) {
            std::printf("deeply nested json: refused (%s)\n", e.what());
        }
    }
    {  // PNG writer on odd sizes
        std::vector<uint8_t> px(37 * 19 * 4);
        for (size_t i = 0; i < px.size(); ++i) px[i] = (uint8_t)(i * 7);
        if (!rto::write_png_rgba8(tmp + "/o.png", px.data(), 37, 19) || !rto::write_png_rgba8(tmp + "/o1.png", px.data(), 1, 1)) rc |= 1;
    }
    std::printf(rc ? "FAILED\n" : "host loaders: clean under ASan + UBSan\n");
    return rc;
}
