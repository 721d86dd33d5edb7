// npz.cpp -- see npz.h.
#include "npz.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace rto {

namespace {

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

[[noreturn]] void fail(const std::string& what) { throw std::runtime_error("npz: " + what); }

// value of `'key': <value>` in the header dict, up to the next top-level comma / brace
std::string dict_value(const std::string& hdr, const std::string& key) {
    size_t k = hdr.find("'" + key + "'");
    if (k == std::string::npos) fail("npy header lacks '" + key + "'");
    size_t c = hdr.find(':', k);
    if (c == std::string::npos) fail("malformed npy header");
    size_t i = c + 1;
    while (i < hdr.size() && hdr[i] == ' ') ++i;
    size_t j = i;
    int depth = 0;
    bool in_str = false;
    for (; j < hdr.size(); ++j) {
        const char ch = hdr[j];
        if (ch == '\'') in_str = !in_str;
        if (in_str) continue;
        if (ch == '(' || ch == '[') ++depth;
        if (ch == ')' || ch == ']') --depth;
        if ((ch == ',' || ch == '}') && depth == 0) break;
    }
    return hdr.substr(i, j - i);
}

}  // namespace

NpyArray parse_npy(const uint8_t* b, size_t n, std::shared_ptr<std::vector<uint8_t>> owner) {
    if (n < 10 || std::memcmp(b, "\x93NUMPY", 6) != 0) fail("bad .npy magic");
    const int major = b[6];
    size_t hlen, hoff;
    if (major == 1) {
        hlen = rd16(b + 8);
        hoff = 10;
    } else if (major == 2 || major == 3) {
        if (n < 12) fail("truncated .npy header");
        hlen = rd32(b + 8);
        hoff = 12;
    } else {
        fail("unsupported .npy version");
    }
    if (hoff + hlen > n) fail("truncated .npy header");
    const std::string hdr(reinterpret_cast<const char*>(b + hoff), hlen);

    NpyArray a;
    std::string descr = dict_value(hdr, "descr");
    if (descr.size() < 2 || descr.front() != '\'') fail("structured dtypes are not supported");
    descr = descr.substr(1, descr.find('\'', 1) - 1);
    a.descr = descr;
    if (descr.size() < 3) fail("bad descr '" + descr + "'");
    if (descr[0] == '>') fail("big-endian arrays are not supported");
    a.kind = descr[1];
    a.word_size = (size_t)std::atoi(descr.c_str() + 2);
    if (a.kind == 'U') a.word_size *= 4;  // UTF-32 code units (cnpy's unicode patch)
    a.fortran_order = dict_value(hdr, "fortran_order").find("True") != std::string::npos;
    std::string shp = dict_value(hdr, "shape");
    for (size_t i = 0; i < shp.size();) {
        if (shp[i] >= '0' && shp[i] <= '9') {
            size_t j = i;
            size_t v = 0;
            while (j < shp.size() && shp[j] >= '0' && shp[j] <= '9') v = v * 10 + (size_t)(shp[j++] - '0');
            a.shape.push_back(v);
            i = j;
        } else {
            ++i;
        }
    }
    a.data = b + hoff + hlen;
    if (a.word_size > 256) fail("bad descr '" + descr + "'");
    {  // element count and byte size with overflow checks (a crafted shape must not wrap around)
        size_t vals = 1;
        for (size_t d : a.shape) {
            if (d != 0 && vals > SIZE_MAX / d) fail("array shape overflows size_t");
            vals *= d;
        }
        if (a.word_size != 0 && vals > SIZE_MAX / a.word_size) fail("array size overflows size_t");
        a.nbytes = vals * a.word_size;
    }
    if (a.nbytes > n - (hoff + hlen)) fail("truncated .npy payload");
    a.owned = std::move(owner);
    // A stored zip member starts wherever its local header ends (np.savez aligns the payload inside the .npy only), so a
    // mapped payload is usually NOT aligned for its element type.  Typed reads of such a pointer are undefined behaviour --
    // and a real fault once a compiler vectorises the reading loop with aligned loads after peeling "up to alignment".
    // Such a member is copied into aligned storage (what the reference's cnpy does for every array); aligned ones stay mapped.
    const size_t align = a.kind == 'U' ? 4 : (a.word_size >= 8 ? 8 : a.word_size >= 4 ? 4 : a.word_size >= 2 ? 2 : 1);
    if (a.nbytes && reinterpret_cast<uintptr_t>(a.data) % align != 0) {
        auto copy = std::make_shared<std::vector<uint8_t>>(a.data, a.data + a.nbytes);
        a.data = copy->data();
        a.owned = std::move(copy);
    }
    return a;
}

NpzFile::~NpzFile() {
    if (map_) munmap(const_cast<uint8_t*>(map_), size_);
    if (fd_ >= 0) close(fd_);
}

const NpyArray& NpzFile::at(const std::string& name) const {
    auto it = arrays_.find(name);
    if (it == arrays_.end()) fail("array '" + name + "' not found");
    return it->second;
}

void NpzFile::open(const std::string& path) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) fail("cannot open '" + path + "'");
    struct stat st;
    if (fstat(fd_, &st) != 0 || st.st_size < 22) fail("'" + path + "' is not a zip file");
    size_ = (size_t)st.st_size;
    void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) fail("mmap failed for '" + path + "'");
    map_ = static_cast<const uint8_t*>(m);

    // End-of-central-directory record (searched backwards over a possible archive comment)
    size_t eocd = std::string::npos;
    const size_t lo = size_ > 22 + 65535 ? size_ - 22 - 65535 : 0;
    for (size_t p = size_ - 22 + 1; p-- > lo;) {
        if (rd32(map_ + p) == 0x06054b50u) {
            eocd = p;
            break;
        }
    }
    if (eocd == std::string::npos) fail("zip end-of-central-directory not found");
    uint64_t n_entries = rd16(map_ + eocd + 10);
    uint64_t cd_size = rd32(map_ + eocd + 12);
    uint64_t cd_off = rd32(map_ + eocd + 16);
    if (n_entries == 0xffff || cd_size == 0xffffffffu || cd_off == 0xffffffffu) {
        // ZIP64: locator sits right before the EOCD
        if (eocd < 20 || rd32(map_ + eocd - 20) != 0x07064b50u) fail("zip64 locator missing");
        const uint64_t e64 = rd64(map_ + eocd - 20 + 8);
        if (e64 > size_ || size_ - e64 < 56 || rd32(map_ + e64) != 0x06064b50u) fail("zip64 EOCD missing");
        n_entries = rd64(map_ + e64 + 32);
        cd_size = rd64(map_ + e64 + 40);
        cd_off = rd64(map_ + e64 + 48);
    }
    if (cd_size > size_ || cd_off > size_ - cd_size) fail("central directory out of range");

    size_t p = (size_t)cd_off;
    for (uint64_t i = 0; i < n_entries; ++i) {
        if (p > size_ || size_ - p < 46 || rd32(map_ + p) != 0x02014b50u) fail("bad central directory entry");
        const uint16_t method = rd16(map_ + p + 10);
        uint64_t csize = rd32(map_ + p + 20), usize = rd32(map_ + p + 24);
        const uint16_t nlen = rd16(map_ + p + 28), xlen = rd16(map_ + p + 30), clen = rd16(map_ + p + 32);
        uint64_t lho = rd32(map_ + p + 42);
        if (size_ - p - 46 < (size_t)nlen + xlen + clen) fail("central directory entry runs past the end of the file");
        std::string name(reinterpret_cast<const char*>(map_ + p + 46), nlen);
        // ZIP64 extended information (header id 1): fields present only for saturated values, in order
        const uint8_t* x = map_ + p + 46 + nlen;
        for (size_t q = 0; q + 4 <= xlen;) {
            const uint16_t id = rd16(x + q), sz = rd16(x + q + 2);
            if (q + 4 + sz > xlen) fail("bad extra field of '" + name + "'");
            if (id == 1) {
                size_t r = q + 4;
                const size_t end = q + 4 + sz;
                auto take = [&](uint64_t& v) {
                    if (r + 8 > end) fail("bad zip64 extra field of '" + name + "'");
                    v = rd64(x + r);
                    r += 8;
                };
                if (usize == 0xffffffffu) take(usize);
                if (csize == 0xffffffffu) take(csize);
                if (lho == 0xffffffffu) take(lho);
            }
            q += 4 + sz;
        }
        p += 46 + (size_t)nlen + xlen + clen;

        if (lho > size_ || size_ - lho < 30 || rd32(map_ + lho) != 0x04034b50u) fail("bad local header for '" + name + "'");
        const size_t doff = (size_t)lho + 30 + rd16(map_ + lho + 26) + rd16(map_ + lho + 28);
        if (doff > size_ || csize > size_ - doff) fail("member '" + name + "' out of range");
        // (deflate expands at most ~1032 : 1: a larger claim is a corrupt size field, not a reason to allocate it)
        if (method == 8 && usize / 1100 > csize + 1) fail("member '" + name + "' claims an impossible inflated size");

        // numpy appends ".npy" to every key
        if (name.size() > 4 && name.compare(name.size() - 4, 4, ".npy") == 0) name.resize(name.size() - 4);

        if (method == 0) {
            arrays_[name] = parse_npy(map_ + doff, (size_t)csize, nullptr);
        } else if (method == 8) {
            auto buf = std::make_shared<std::vector<uint8_t>>((size_t)usize);
            z_stream zs;
            std::memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) fail("inflateInit2 failed");
            // feed in < 4 GiB pieces (z_stream counters are 32-bit)
            size_t in_done = 0, out_done = 0;
            int rc = Z_OK;
            while (rc != Z_STREAM_END) {
                const size_t in_chunk = std::min<size_t>((size_t)csize - in_done, 1u << 30);
                const size_t out_chunk = std::min<size_t>((size_t)usize - out_done, 1u << 30);
                zs.next_in = const_cast<Bytef*>(map_ + doff + in_done);
                zs.avail_in = (uInt)in_chunk;
                zs.next_out = buf->data() + out_done;
                zs.avail_out = (uInt)out_chunk;
                rc = inflate(&zs, Z_NO_FLUSH);
                in_done += in_chunk - zs.avail_in;
                out_done += out_chunk - zs.avail_out;
                if (rc != Z_OK && rc != Z_STREAM_END) {
                    inflateEnd(&zs);
                    fail("inflate failed for '" + name + "'");
                }
                if (rc == Z_OK && in_chunk == 0 && out_chunk == 0) break;
            }
            inflateEnd(&zs);
            if (out_done != usize) fail("short inflate for '" + name + "'");
            arrays_[name] = parse_npy(buf->data(), buf->size(), buf);
        } else {
            fail("unsupported zip compression method for '" + name + "'");
        }
    }
}

NpyArray load_npy_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot open '" + path + "'");
    auto buf = std::make_shared<std::vector<uint8_t>>((std::istreambuf_iterator<char>(f)),
                                                       std::istreambuf_iterator<char>());
    return parse_npy(buf->data(), buf->size(), buf);
}

}  // namespace rto
