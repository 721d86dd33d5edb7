// n3tree_host.h -- host-side PlenOctree ("N3Tree") loader: tree.npz -> arrays ready for upload.
// Reference: renderer/src/n3tree.cpp:55-78 (DataFormat::parse), :111-154 (N3Tree::open),
// :228-362 (load_npz incl. the quantised decode :279-340), :20-53 (LLFF poses_bounds side file).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "npz.h"

namespace rto {

struct DataFormat {  // data_format.hpp:7-25
    int format = 0;  // RTO_FMT_*: RGBA, SH, SG, ASG
    int basis_dim = -1;
    void parse(const std::string& s);  // n3tree.cpp:55-78
    std::string to_string() const;     // n3tree.cpp:80-101
};

struct HostTree {
    int N = 0;
    int64_t capacity = 0;
    int data_dim = 0;
    DataFormat data_format;
    float scale[3] = {1, 1, 1};
    float offset[3] = {0, 0, 0};
    const int32_t* child = nullptr;   // [capacity*N^3]
    const uint16_t* data = nullptr;   // fp16 bits [capacity*N^3*data_dim]
    // LLFF / NDC (n3tree.cpp:131-148)
    bool use_ndc = false;
    float ndc_width = 0, ndc_height = 0, ndc_focal = 0;
    float ndc_avg_up[3] = {0, 0, 0}, ndc_avg_back[3] = {0, 0, 0}, ndc_avg_cen[3] = {0, 0, 0};

    // quantised set kept as stored (open(path, /*keep_quantized=*/true)): `data` stays nullptr
    bool quantized = false;
    int n_basis = 0, n_retain = 0;
    const uint16_t* q_map = nullptr;       // u16  [n_basis - n_retain][capacity*N^3]
    const uint16_t* q_colors = nullptr;    // fp16 [n_basis - n_retain][65536][3]
    const uint16_t* q_sigma = nullptr;     // fp16 [capacity*N^3]
    const uint16_t* q_retained = nullptr;  // fp16 [n_retain][capacity*N^3][3]

    // keep-alives
    std::shared_ptr<NpzFile> npz;
    std::vector<uint16_t> decoded;  // quantised trees are expanded here

    // throws std::runtime_error (bad dtype / schema), returns false when the file does not exist
    // (the reference prints a message and leaves the tree empty, n3tree.cpp:123-126)
    bool open(const std::string& path, bool keep_quantized = false);
};

// deepest leaf level (number of child[] loads to reach it), validating every offset on the way;
// throws std::runtime_error on an out-of-range child
int tree_max_depth(const int32_t* child, int64_t capacity, int N);

}  // namespace rto
