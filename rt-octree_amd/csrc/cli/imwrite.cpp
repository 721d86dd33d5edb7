#include "imwrite.h"

#include <zlib.h>

#include <cstdio>
#include <vector>

namespace rto {
namespace {
void put32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}
void chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    put32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
}
}  // namespace

bool write_png_rgba8(const std::string& path, const uint8_t* rgba, int width, int height) {
    if (width <= 0 || height <= 0 || !rgba) return false;
    // scanlines: filter byte 0 (PNG_FILTER_NONE, imwrite.cpp:58) + 4*width bytes
    const size_t stride = (size_t)width * 4;
    std::vector<uint8_t> raw((stride + 1) * (size_t)height);
    for (int y = 0; y < height; ++y) {
        raw[(stride + 1) * y] = 0;
        std::copy(rgba + stride * y, rgba + stride * (y + 1), raw.begin() + (stride + 1) * y + 1);
    }
    // zlib stream with stored blocks (compression level 0, imwrite.cpp:57)
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 0) != Z_OK) return false;
    z.resize(zlen);

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)width);
    put32(ihdr, (uint32_t)height);
    ihdr.push_back(8);  // bit depth
    ihdr.push_back(6);  // colour type RGBA (imwrite.cpp:51-55)
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    return std::fclose(f) == 0 && ok;
}
}  // namespace rto
