// n3tree_host.cpp -- see n3tree_host.h.
#include "n3tree_host.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

#include "rto.h"

namespace rto {

// n3tree.cpp:55-78: leading letters = format name, trailing integer = basis_dim
void DataFormat::parse(const std::string& str) {
    size_t nonalph = std::string::npos;
    for (size_t i = 0; i < str.size(); ++i) {
        if (!std::isalpha((unsigned char)str[i])) {
            nonalph = i;
            break;
        }
    }
    if (nonalph != std::string::npos) {
        basis_dim = std::atoi(str.c_str() + nonalph);
        const std::string tmp = str.substr(0, nonalph);
        if (tmp == "ASG")
            format = RTO_FMT_ASG;
        else if (tmp == "SG")
            format = RTO_FMT_SG;
        else if (tmp == "SH")
            format = RTO_FMT_SH;
        else
            format = RTO_FMT_RGBA;
    } else {
        basis_dim = -1;
        format = RTO_FMT_RGBA;
    }
}

std::string DataFormat::to_string() const {  // n3tree.cpp:80-101
    std::string out;
    switch (format) {
        case RTO_FMT_ASG: out = "ASG"; break;
        case RTO_FMT_SG: out = "SG"; break;
        case RTO_FMT_SH: out = "SH"; break;
        case RTO_FMT_RGBA: out = "RGBA"; break;
        default: out = "UNKNOWN";
    }
    if (basis_dim != -1) out.append(std::to_string(basis_dim));
    return out;
}

namespace {

double scalar_as_double(const NpyArray& a, size_t i) {
    if (i >= a.num_vals()) throw std::runtime_error("tree.npz: array of " + std::to_string(a.num_vals()) + " values read at index " + std::to_string(i));
    if (a.kind == 'f' && a.word_size == 4) return a.as<float>()[i];
    if (a.kind == 'f' && a.word_size == 8) return a.as<double>()[i];
    if (a.kind == 'i' && a.word_size == 8) return (double)a.as<int64_t>()[i];
    if (a.kind == 'i' && a.word_size == 4) return (double)a.as<int32_t>()[i];
    if (a.kind == 'u' && a.word_size == 8) return (double)a.as<uint64_t>()[i];
    if (a.kind == 'u' && a.word_size == 4) return (double)a.as<uint32_t>()[i];
    throw std::runtime_error("tree.npz: unsupported scalar dtype " + a.descr);
}

// n3tree.cpp:20-53 unpack_llff_poses_bounds (17 values per camera: 3x5 pose + 2 bounds)
void unpack_llff(const NpyArray& pb, HostTree& t) {
    const size_t n = pb.num_vals();
    auto v = [&](size_t i) { return (float)scalar_as_double(pb, i); };
    t.ndc_height = v(4);
    t.ndc_width = v(9);
    t.ndc_focal = v(14);
    float cen[3] = {0, 0, 0}, back[3] = {0, 0, 0}, right[3] = {0, 0, 0}, up[3] = {0, 0, 0};
    const size_t BLOCK = 17;
    float bd_min = 1e9f;
    for (size_t off = 0; off + BLOCK <= n; off += BLOCK) {
        for (size_t r = 0; r < 3; ++r) {
            right[r] += v(off + 5 * r + 1);
            up[r] -= v(off + 5 * r + 0);
            back[r] += v(off + 5 * r + 2);
            cen[r] += v(off + 5 * r + 3);
        }
        bd_min = std::fmin(bd_min, std::fmin(v(off + 15), v(off + 16)));
    }
    const float total = (float)(n / BLOCK);
    auto normalize = [](float* a) {
        const float l = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        for (int i = 0; i < 3; ++i) a[i] /= l;
    };
    auto cross = [](const float* a, const float* b, float* o) {
        o[0] = a[1] * b[2] - a[2] * b[1];
        o[1] = a[2] * b[0] - a[0] * b[2];
        o[2] = a[0] * b[1] - a[1] * b[0];
    };
    for (int i = 0; i < 3; ++i) cen[i] = cen[i] / (total * bd_min * 0.75f);
    normalize(back);
    cross(up, back, right);
    normalize(right);
    cross(back, right, up);
    normalize(up);
    for (int i = 0; i < 3; ++i) {
        t.ndc_avg_cen[i] = cen[i];
        t.ndc_avg_back[i] = back[i];
        t.ndc_avg_up[i] = up[i];
    }
}

}  // namespace

bool HostTree::open(const std::string& path, bool keep_quantized) {
    if (path.size() <= 4 || path.substr(path.size() - 4) != ".npz")  // n3tree.cpp:119 (assert)
        throw std::runtime_error("tree file must end in .npz: " + path);
    if (!std::ifstream(path)) {
        std::fprintf(stderr, "Can't load because file does not exist: %s\n", path.c_str());
        return false;
    }
    npz = std::make_shared<NpzFile>();
    npz->open(path);
    const NpzFile& z = *npz;

    // ---- load_npz n3tree.cpp:228-362 ----
    data_dim = (int)scalar_as_double(z.at("data_dim"), 0);
    if (z.has("data_format")) {
        const NpyArray& df = z.at("data_format");
        std::string s;
        if (df.kind == 'U') {  // UTF-32 -> ASCII (n3tree.cpp:231-238)
            for (size_t i = 0; i + 4 <= df.nbytes; i += 4)
                if (df.data[i]) s.push_back((char)df.data[i]);
        } else {  // 'S' bytes
            for (size_t i = 0; i < df.nbytes; ++i)
                if (df.data[i]) s.push_back((char)df.data[i]);
        }
        data_format.parse(s);
    } else if (data_dim == 4) {
        data_format.format = RTO_FMT_RGBA;
        data_format.basis_dim = -1;
        std::fprintf(stderr, "INFO: Legacy file with no format specifier; spherical basis disabled\n");
    } else {
        data_format.format = RTO_FMT_SH;
        data_format.basis_dim = (data_dim - 1) / 3;
        std::fprintf(stderr, "INFO: Legacy file with no format specifier; autodetect spherical harmonics order\n");
    }
    std::fprintf(stderr, "INFO: Data format %s\n", data_format.to_string().c_str());

    if (z.has("invradius3")) {
        const NpyArray& a = z.at("invradius3");
        if (a.num_vals() < 3) throw std::runtime_error("tree.npz: invradius3 must have 3 values");
        for (int i = 0; i < 3; ++i) scale[i] = (float)scalar_as_double(a, i);
    } else {
        scale[0] = scale[1] = scale[2] = (float)scalar_as_double(z.at("invradius"), 0);
    }
    {
        const NpyArray& a = z.at("offset");
        if (a.num_vals() < 3) throw std::runtime_error("tree.npz: offset must have 3 values");
        for (int i = 0; i < 3; ++i) offset[i] = (float)scalar_as_double(a, i);
    }

    const NpyArray& ch = z.at("child");
    if (!(ch.kind == 'i' && ch.word_size == 4) || ch.shape.size() != 4 || ch.fortran_order)
        throw std::runtime_error("tree.npz: child must be a C-ordered int32 [capacity,N,N,N] array");
    N = (int)ch.shape[1];
    if (N < 1 || ch.shape[2] != ch.shape[1] || ch.shape[3] != ch.shape[1])
        throw std::runtime_error("tree.npz: child must be [capacity,N,N,N] with equal N");
    if (N != 2) std::fprintf(stderr, "WARNING: N != 2 probably doesn't work.\n");
    child = ch.as<int32_t>();
    const size_t N3 = (size_t)N * N * N;

    if (z.has("quant_colors")) {  // n3tree.cpp:279-340
        std::fprintf(stderr, "INFO: Decoding quantized colors\n");
        const NpyArray& qc = z.at("quant_colors");
        if (qc.word_size != 2) throw std::runtime_error("codebook must be stored in half precision");
        if (qc.shape.size() < 2) throw std::runtime_error("tree.npz: quant_colors must be [n_quantised,65536,3]");
        const NpyArray& qm = z.at("quant_map");
        if (qm.word_size != 2 || qm.shape.size() < 2) throw std::runtime_error("tree.npz: quant_map must be uint16");
        capacity = (int64_t)qm.shape[1];
        int n_basis = (int)qm.shape[0];
        if ((int)qc.shape[0] != n_basis) throw std::runtime_error("codebook and map basis numbers does not match");
        if (z.has("data_retained") && z.at("data_retained").shape.empty())
            throw std::runtime_error("tree.npz: data_retained must be fp16 [n_retain,capacity,N,N,N,3]");
        const int n_retain = z.has("data_retained") ? (int)z.at("data_retained").shape[0] : 0;
        n_basis += n_retain;
        const NpyArray& sg = z.at("sigma");
        if (sg.word_size != 2) throw std::runtime_error("tree.npz: sigma must be stored in half precision");
        const size_t n_child = (size_t)capacity * N3;
        if (sg.num_vals() < n_child || qm.num_vals() < (size_t)(n_basis - n_retain) * n_child)
            throw std::runtime_error("tree.npz: quantised arrays are too small for the tree");
        if (data_dim < 3 * n_basis + 1) throw std::runtime_error("tree.npz: data_dim too small for the quantised bases");
        // both branches below index the codebook with 16-bit ids and data_retained per slot
        if (qc.num_vals() < (size_t)(n_basis - n_retain) * 65536 * 3)
            throw std::runtime_error("tree.npz: quant_colors must be [n_quantised,65536,3]");
        if (n_retain) {
            const NpyArray& rt = z.at("data_retained");
            if (rt.word_size != 2 || rt.num_vals() < (size_t)n_retain * n_child * 3)
                throw std::runtime_error("tree.npz: data_retained must be fp16 [n_retain,capacity,N,N,N,3]");
        }
        if (keep_quantized) {  // render straight from the codebooks: nothing is expanded
            quantized = true;
            this->n_basis = n_basis;
            this->n_retain = n_retain;
            q_map = qm.as<uint16_t>();
            q_colors = qc.as<uint16_t>();
            q_sigma = sg.as<uint16_t>();
            if (n_retain) q_retained = z.at("data_retained").as<uint16_t>();
            data = nullptr;
        } else {
        decoded.assign(n_child * (size_t)data_dim, 0);
        const uint16_t* sigma = sg.as<uint16_t>();
        const uint16_t* qmap = qm.as<uint16_t>();
        const uint16_t* qcol = qc.as<uint16_t>();
        uint16_t* out = decoded.data();
        for (size_t i = 0; i < n_child; ++i) {
            const size_t off = i * (size_t)data_dim;
            for (int j = 0; j < n_basis - n_retain; ++j) {
                size_t boff = off + (size_t)j + (size_t)n_retain;
                const size_t id = qmap[(size_t)j * n_child + i];
                const uint16_t* col = qcol + (size_t)j * 65536 * 3 + id * 3;
                for (int k = 0; k < 3; ++k) {
                    out[boff] = col[k];
                    boff += (size_t)n_basis;
                }
            }
            out[off + (size_t)data_dim - 1] = sigma[i];
        }
        if (n_retain) {
            const uint16_t* rp = z.at("data_retained").as<uint16_t>();
            for (size_t i = 0; i < n_child; ++i) {
                const size_t off = i * (size_t)data_dim;
                for (int j = 0; j < n_retain; ++j) {
                    size_t boff = off + (size_t)j;
                    const uint16_t* col = rp + (size_t)j * n_child * 3 + i * 3;
                    for (int k = 0; k < 3; ++k) {
                        out[boff] = col[k];
                        boff += (size_t)n_basis;
                    }
                }
            }
        }
        data = decoded.data();
        }
    } else {
        const NpyArray& d = z.at("data");
        if (d.shape.empty()) throw std::runtime_error("tree.npz: data must be [capacity,N,N,N,data_dim]");
        capacity = (int64_t)d.shape[0];
        if (d.word_size != 2) throw std::runtime_error("data must be stored in half precision");
        if (d.fortran_order) throw std::runtime_error("tree.npz: data must be C-ordered");
        if (d.num_vals() != (size_t)capacity * N3 * (size_t)data_dim)
            throw std::runtime_error("tree.npz: data shape does not match capacity*N^3*data_dim");
        data = d.as<uint16_t>();
    }
    if ((int64_t)ch.shape[0] < capacity) throw std::runtime_error("tree.npz: child has fewer nodes than data");

    // ---- NDC side file n3tree.cpp:131-148 ----
    const std::string pb_path = path.substr(0, path.size() - 4) + "_poses_bounds.npy";
    use_ndc = bool(std::ifstream(pb_path));
    if (use_ndc) {
        std::fprintf(stderr, "INFO: Found poses_bounds.npy for NDC: %s\n", pb_path.c_str());
        NpyArray pb = load_npy_file(pb_path);
        unpack_llff(pb, *this);
    }
    return true;
}

int tree_max_depth(const int32_t* child, int64_t capacity, int N) {
    const int64_t N3 = (int64_t)N * N * N;
    if (capacity <= 0) return 0;
    std::vector<std::pair<int64_t, int>> stack;
    stack.emplace_back(0, 1);
    int maxd = 0;
    int64_t visited = 0;
    while (!stack.empty()) {
        const auto [node, lvl] = stack.back();
        stack.pop_back();
        if (++visited > capacity) throw std::runtime_error("tree: child offsets form a cycle");
        if (lvl > maxd) maxd = lvl;
        const int32_t* c = child + node * N3;
        for (int64_t s = 0; s < N3; ++s) {
            if (c[s] != 0) {
                const int64_t nxt = node + c[s];
                if (nxt < 0 || nxt >= capacity) throw std::runtime_error("tree: child offset out of range");
                stack.emplace_back(nxt, lvl + 1);
            }
        }
    }
    return maxd;
}

}  // namespace rto
