// imwrite.h -- RGBA8 PNG writer (the role of renderer/src/imwrite.cpp:14-86, which uses libpng with
// compression level 0 and no filter).  png.h is not installed in this image, so the container is
// written by hand: IHDR / one IDAT holding a zlib stream of stored (uncompressed) deflate blocks /
// IEND, CRCs from zlib.  Pixel bytes after decoding are identical to the reference's files.
#pragma once
#include <cstdint>
#include <string>

namespace rto {
// returns false on I/O failure
bool write_png_rgba8(const std::string& path, const uint8_t* rgba, int width, int height);
}  // namespace rto
