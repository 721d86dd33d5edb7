// npz.h -- read-only .npy/.npz access (the role of the reference's vendored cnpy,
// renderer/3rdparty/cnpy/cnpy.cpp:67-175 parse_npy_header, :227-263 load_the_npz_array,
// :303-369 npz_load, including its ZIP64 and '<U' string patches).
//
// Design: the file is mmap()ed and indexed through the ZIP central directory, so STORED members
// (what numpy.savez writes) are exposed zero-copy -- a 2 GB tree.npz goes from page cache to
// hipMemcpy without an intermediate std::vector; DEFLATE members (savez_compressed) are inflated
// with zlib into owned buffers.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace rto {

struct NpyArray {
    std::string descr;           // e.g. "<f2", "<i4", "<U4", "|u1"
    char kind = 0;               // 'f','i','u','U','b', ...
    size_t word_size = 0;        // bytes per element
    bool fortran_order = false;
    std::vector<size_t> shape;
    const uint8_t* data = nullptr;  // element bytes (points into the mmap or into `owned`)
    size_t nbytes = 0;
    std::shared_ptr<std::vector<uint8_t>> owned;  // set for inflated members

    size_t num_vals() const {
        size_t n = 1;
        for (size_t s : shape) n *= s;
        return n;
    }
    template <typename T>
    const T* as() const {
        return reinterpret_cast<const T*>(data);
    }
};

class NpzFile {
public:
    NpzFile() = default;
    ~NpzFile();
    NpzFile(const NpzFile&) = delete;
    NpzFile& operator=(const NpzFile&) = delete;

    // throws std::runtime_error
    void open(const std::string& path);
    bool has(const std::string& name) const { return arrays_.count(name) != 0; }
    const NpyArray& at(const std::string& name) const;
    const std::map<std::string, NpyArray>& arrays() const { return arrays_; }

private:
    int fd_ = -1;
    const uint8_t* map_ = nullptr;
    size_t size_ = 0;
    std::map<std::string, NpyArray> arrays_;
};

// A bare .npy file (LLFF poses_bounds.npy: n3tree.cpp:131-148, main_headless.cpp:300)
NpyArray load_npy_file(const std::string& path);

// parse a .npy image held in memory; `owner` keeps the bytes alive when non-null
NpyArray parse_npy(const uint8_t* bytes, size_t n, std::shared_ptr<std::vector<uint8_t>> owner);

}  // namespace rto
