// poses.h -- camera pose loaders of volrend_headless (renderer/main_headless.cpp:64-190,255-390):
// blender transforms_*.json, TanksAndTemple pose directories (+ ../intrinsics.txt) and LLFF
// poses_bounds.npy, plus the camera-convention fix-ups applied after loading.
#pragma once
#include <array>
#include <string>
#include <vector>

namespace rto {

// 4x3 column-major camera-to-world (glm::mat4x3): m[c*3 + r], columns 0..2 axes, column 3 centre
using Mat4x3 = std::array<float, 12>;

struct PoseSet {
    std::vector<Mat4x3> trans;
    std::vector<std::string> basenames;
    int width = 800, height = 800;
    float fx = 1111.11f, fy = 1111.11f;
    bool is_llff = false;
};

// throws std::runtime_error; `dataset` in {"blender","tt","llff"}; width/height/fx/fy in `ps` hold
// the command-line values on entry (opts.cpp:15-20) and the dataset's values on return
void load_poses(const std::string& dataset, const std::string& poses_path, bool reverse_yz, PoseSet& ps);

// m := m * diag(1,-1,-1,1) (OpenCV -> NeRF convention, main_headless.cpp:373-384)
void flip_yz(Mat4x3& m);

}  // namespace rto
