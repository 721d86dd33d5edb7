#include "poses.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "../host/mini_json.h"
#include "../host/npz.h"

namespace fs = std::filesystem;

namespace rto {
namespace {

struct V3 {
    float x, y, z;
};
V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
V3 normalize(V3 v) {  // main_headless.cpp:106-110
    const float n = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return {v.x / n, v.y / n, v.z / n};
}
V3 col(const Mat4x3& m, int c) { return {m[c * 3], m[c * 3 + 1], m[c * 3 + 2]}; }
void set_col(Mat4x3& m, int c, V3 v) {
    m[c * 3] = v.x;
    m[c * 3 + 1] = v.y;
    m[c * 3 + 2] = v.z;
}

std::string remove_ext(const std::string& s) {  // main_headless.cpp:51-62
    const size_t i = s.rfind('.');
    return i == std::string::npos ? s : s.substr(0, i);
}

// main_headless.cpp:64-92: whitespace-separated 4x4 (or 3x4) row-major matrices, several per file
int read_transform_matrices(const std::string& path, std::vector<Mat4x3>& out) {
    std::ifstream ifs(path);
    if (!ifs) throw std::runtime_error("'" + path + "' does not exist");
    int cnt = 0;
    while (ifs) {
        Mat4x3 m{};
        float g;
        float r[3][4];
        ifs >> r[0][0] >> r[0][1] >> r[0][2] >> r[0][3];
        if (!ifs) break;
        ifs >> r[1][0] >> r[1][1] >> r[1][2] >> r[1][3];
        ifs >> r[2][0] >> r[2][1] >> r[2][2] >> r[2][3];
        if (ifs) ifs >> g >> g >> g >> g;
        for (int c = 0; c < 4; ++c)
            for (int i = 0; i < 3; ++i) m[c * 3 + i] = r[i][c];
        ++cnt;
        out.push_back(m);
    }
    return cnt;
}

void read_intrins(const std::string& path, float& fx, float& fy) {  // main_headless.cpp:94-105
    std::ifstream ifs(path);
    if (!ifs) throw std::runtime_error("intrin '" + path + "' does not exist");
    float _;
    ifs >> fx >> _ >> _ >> _;
    ifs >> _ >> fy;
}

// 4x4 helpers for the LLFF recentring (main_headless.cpp:147-188); column-major, m[c*4 + r]
using M4 = std::array<float, 16>;
M4 expand(const Mat4x3& a) {
    M4 r{};
    for (int c = 0; c < 4; ++c) {
        for (int i = 0; i < 3; ++i) r[c * 4 + i] = a[c * 3 + i];
        r[c * 4 + 3] = c == 3 ? 1.f : 0.f;
    }
    return r;
}
M4 mul(const M4& a, const M4& b) {
    M4 r{};
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 4; ++i) {
            float s = 0;
            for (int k = 0; k < 4; ++k) s += a[k * 4 + i] * b[c * 4 + k];
            r[c * 4 + i] = s;
        }
    return r;
}
// inverse of a rigid-ish 4x4 by Gauss-Jordan (glm::inverse in the reference, :181)
M4 inverse(const M4& m) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = m[j * 4 + i];
            a[i][j + 4] = i == j ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r)
            if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
        if (std::fabs(a[p][c]) < 1e-30) throw std::runtime_error("singular average pose");
        if (p != c)
            for (int j = 0; j < 8; ++j) std::swap(a[p][j], a[c][j]);
        const double d = a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] /= d;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = a[r][c];
                for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
            }
    }
    M4 r{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r[j * 4 + i] = (float)a[i][j + 4];
    return r;
}

Mat4x3 viewmatrix(V3 z, V3 up, V3 pos) {  // main_headless.cpp:147-154
    z = normalize(z);
    const V3 x = normalize(cross(up, z));
    const V3 y = normalize(cross(z, x));
    Mat4x3 m{};
    set_col(m, 0, x);
    set_col(m, 1, y);
    set_col(m, 2, z);
    set_col(m, 3, pos);
    return m;
}

void recenter_poses(std::vector<Mat4x3>& trans) {  // main_headless.cpp:156-188
    V3 z{0, 0, 0}, up{0, 0, 0}, cen{0, 0, 0};
    for (const auto& t : trans) {
        z = z + col(t, 2);
        up = up + col(t, 1);
        cen = cen + col(t, 3);
    }
    const float n = (float)trans.size();
    const Mat4x3 avg = viewmatrix(normalize(z / n), up / n, cen / n);
    const M4 inv = inverse(expand(avg));
    for (auto& t : trans) {
        const M4 p = mul(inv, expand(t));
        for (int c = 0; c < 4; ++c)
            for (int i = 0; i < 3; ++i) t[c * 3 + i] = p[c * 4 + i];
    }
}

}  // namespace

void flip_yz(Mat4x3& m) {
    for (int i = 0; i < 3; ++i) {
        m[3 + i] = -m[3 + i];
        m[6 + i] = -m[6 + i];
    }
}

void load_poses(const std::string& dataset, const std::string& poses_path, bool reverse_yz, PoseSet& ps) {
    if (dataset == "blender") {  // main_headless.cpp:255-272
        std::ifstream f(poses_path);
        if (!f) throw std::runtime_error("cannot open poses file '" + poses_path + "'");
        std::stringstream ss;
        ss << f.rdbuf();
        const json::ValuePtr j = json::parse(ss.str());
        const float camera_angle_x = (float)j->at("camera_angle_x").as_number();
        ps.fx = ps.fy = 0.5f * ps.width / std::tan(0.5f * camera_angle_x);
        const json::Value& frames = j->at("frames");
        for (size_t i = 0; i < frames.size(); ++i) {
            const json::Value& m = frames.at(i).at("transform_matrix");
            Mat4x3 t{};
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) t[c * 3 + r] = (float)m.at(r).at(c).as_number();  // transpose
            ps.trans.push_back(t);
            ps.basenames.push_back("r_" + std::to_string(i));
        }
    } else if (dataset == "tt") {  // :273-297
        ps.width = 1920;
        ps.height = 1080;
        read_intrins((fs::path(poses_path) / ".." / "intrinsics.txt").string(), ps.fx, ps.fy);
        std::vector<fs::path> files;
        for (const auto& e : fs::directory_iterator(poses_path)) files.push_back(e.path());
        std::sort(files.begin(), files.end());  // the reference iterates in filesystem order (unsorted)
        for (const auto& p : files) {
            const int cnt = read_transform_matrices(p.string(), ps.trans);
            const std::string fname = remove_ext(p.filename().string());
            if (cnt == 1) {
                ps.basenames.push_back(fname);
            } else {
                for (int i = 0; i < cnt; ++i) {
                    std::string tmp = std::to_string(i);
                    while (tmp.size() < 6) tmp = "0" + tmp;
                    ps.basenames.push_back(fname + "_" + tmp);
                }
            }
        }
    } else if (dataset == "llff") {  // :298-370
        ps.is_llff = true;
        const NpyArray a = load_npy_file(poses_path);
        if (a.shape.size() != 2 || a.shape[1] < 17) throw std::runtime_error("poses_bounds.npy must be [n,17]");
        const size_t n = a.shape[0], stride = a.shape[1];
        auto get = [&](size_t set, size_t k) -> float {
            return a.word_size == 4 ? a.as<float>()[set * stride + k] : (float)a.as<double>()[set * stride + k];
        };
        constexpr int factor = 4;
        ps.width = (int)(get(0, 9) / factor);
        ps.height = (int)(get(0, 4) / factor);
        ps.fx = ps.fy = get(0, 14) / factor;
        float bds_min = 1e9f;
        for (size_t i = 0; i < n; ++i) bds_min = std::min(bds_min, get(i, 15));
        for (size_t i = 0; i < n; ++i) {
            Mat4x3 t{};
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) t[c * 3 + r] = get(i, (size_t)r * 5 + c);
            // temp * cam_trans with cam_trans = [[0,1,0,0],[-1,0,0,0],[0,0,1,0],[0,0,0,1]] (columns): the
            // new column 0 is old column 1, the new column 1 is minus old column 0 (:325-343)
            const V3 c0 = col(t, 0), c1 = col(t, 1);
            set_col(t, 0, c1);
            set_col(t, 1, {-c0.x, -c0.y, -c0.z});
            const float scale = 1.0f / (bds_min * 0.75f);
            for (int r = 0; r < 3; ++r) t[9 + r] *= scale;
            ps.trans.push_back(t);
        }
        std::string images = "images";
        if (factor > 1) images += "_" + std::to_string(factor);
        const fs::path dir = fs::path(poses_path).parent_path() / images;
        if (fs::is_directory(dir)) {
            for (const auto& e : fs::directory_iterator(dir)) ps.basenames.push_back(remove_ext(e.path().filename().string()));
            std::sort(ps.basenames.begin(), ps.basenames.end());
        }
        while (ps.basenames.size() < ps.trans.size()) ps.basenames.push_back("r_" + std::to_string(ps.basenames.size()));
    } else {
        throw std::runtime_error("unknown dataset type '" + dataset + "' (blender|tt|llff)");
    }

    if (dataset == "tt" || reverse_yz) {  // :372-384
        std::puts("INFO: Use OpenCV camera convention\n");
        for (auto& t : ps.trans) flip_yz(t);
    } else if (dataset == "llff") {
        std::puts("INFO: Use LLFF camera convention\n");
        recenter_poses(ps.trans);
    } else {
        std::puts("INFO: Use NeRF camera convention\n");
    }
}

}  // namespace rto
