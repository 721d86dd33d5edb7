// rto_abi.cpp -- implementation of the C ABI declared in include/rto.h.
// Host logic only; every computation on frame data happens in the gfx950 kernels
// (render_kernels.hip, filter_kernels.hip).  There is no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "host/mini_json.h"
#include "host/n3tree_host.h"
#include "rto.h"
#include "rto_launch.h"
#include "rto_denoise_launch.h"

namespace {

constexpr int kKtRing = 256;

thread_local std::string g_err;

int set_err(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return set_err(RTO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                                          std::to_string(__LINE__) + ")");                                   \
    } while (0)

// set the device for the scope of one ABI call, restore the caller's device afterwards
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) {
            ok = false;
            return;
        }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// The raw filtering entry points take device pointers only: launch on the device that owns the output
// buffer (a process that drives several GPUs may have another one current).
int device_of(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return a.device;
}

// a hit-list entry packs {slot, count - 1}: at this SPP the tree's leaf slots must fit hit_slot_bits(spp) bits
// (strictly: the all-ones entry terminates a list)
bool slots_fit_spp(int64_t n_slots, int spp) { return n_slots < (int64_t(1) << rto::hit_slot_bits(spp)); }
// a hit entry names a leaf slot -- or, inside the batched traversal, an entry of the two-level image (a slightly larger index
// space: TreeDev::wide_entries + the slots above the grid levels): both must fit the entry's slot bits at this SPP
bool tree_fits_spp(const rto_tree* tree, int spp);
// How the fast / batched kernels render `tree` at this SPP: 0 = not at all (generic kernel), 1 = with tree->dev as it is,
// 2 = with *td = tree->dev minus the two-level image (its entries do not fit a hit entry at this SPP, the leaf slots do:
// the one-level walk, whose hits name slots).  wide_bits > 0: test hook, pretend a hit entry leaves that many bits for an
// entry of the two-level image.
int fast_path_for_spp(const rto_tree* tree, int spp, int wide_bits, rto::TreeDev* td);

constexpr int kMaxSpp = 32;  // the largest of the supported set below: a hit entry's slot field is narrowest there
bool spp_supported(int spp) {  // volrend.cu:266-278
    return spp == 1 || spp == 2 || spp == 3 || spp == 4 || spp == 6 || spp == 8 || spp == 16 || spp == 32;
}

}  // namespace

struct rto_tree {
    int device = 0;
    rto::TreeDev dev{};
    rto_tree_info info{};
    void* d_data = nullptr;
    void* d_child = nullptr;
    void* d_nodew = nullptr;
    void* d_topgrid = nullptr;
    void *d_widew = nullptr, *d_worig = nullptr, *d_grid2 = nullptr;  // two-level traversal image (build_wide_image)
    void* d_shrec = nullptr;  // aligned copy of the SH coefficients (shading)
    void* d_recidx = nullptr; // RTO_TREE_COMPACT_RECORDS: slot -> record of d_shrec
    void* d_occ = nullptr;    // culling cells (TreeDev::occ_cells)
    void* d_qrec = nullptr;
    void* d_qcolors = nullptr;
    void* d_qsigma = nullptr;
    bool quant = false;  // rendered from the codebooks; only the batched path can shade it
    bool fast_ok = false;
    // Footprint (VERDICT r2 task 8): a dense SH9 / SH16 tree that renders through the fast / batched kernels reads only
    // nodew + topgrid + shrec, so child[] / data[] are released after the upload and rebuilt from those two on the first
    // launch that selects the generic kernel (ensure_reference_arrays).  RTO_TREE_KEEP_REFERENCE keeps them resident.
    std::atomic<bool> reference_dropped{false};  // (read outside rebuild_mutex: the usual double-checked test)
    int shrec_halves = 0;
    std::mutex rebuild_mutex;
};

// tile rows per band of the XCD ray queues (build_tile_tables; 0: angular wedges, rounds 3-5).  Per 100 C2 frames: wedges 3.84 ms,
// bands of 1 / 2 / 4 / 8 tile rows 3.73 / 3.70 / 3.72 / 3.75; through bench.py 3.92 -> 3.78 (2) / 3.75 (3); C5 2.48 -> 2.37, C3 shape
// 3.30 -> 3.15, C4 unchanged (profiles/r6_y_ab_queue_bands*.txt)
#ifndef RTO_QUEUE_BANDS_DEFAULT
#define RTO_QUEUE_BANDS_DEFAULT 3
#endif
struct rto_ctx {
    int device = 0;
    int width = 0, height = 0;
    int frames = 1;  // frame slots (batched launches render slots 0..n-1)
    int sel = 0;     // slot the single-frame entry points and accessors refer to
    int num_cus = 256;
    unsigned long long* queue = nullptr;  // persistent-kernel ray queues (kQueueWords u64)
    rto::FrameDesc* d_frames = nullptr;   // frame table of the batch in flight (kMaxBatch descriptors)
    rto::OccupancyCache occ;              // of the persistent kernel instantiation last launched
    uint32_t* tile_order = nullptr;       // centre-out order of the 8x8 ray tiles (persistent kernel)
    uint32_t* wedge_order = nullptr;      // the same tiles grouped into 8 angular wedges (one ray queue per XCD)
    int wedge_start[rto::kMaxQueues + 1] = {0};
    bool xcd_queues = true;
    bool tile_major = true;
    int tile_block = 4;  // tiles per side of the blocks the wedge queues are ordered by
    int queue_bands = RTO_QUEUE_BANDS_DEFAULT;  // > 0: the XCD queues take BANDS of this many tile rows (band j -> queue j % 8) instead of angular wedges
    uint32_t* hits = nullptr;             // [frames][hits_spp][H*W] traversal -> shading hand-off
    int hits_spp = 0;
    // empty-space culling + ray-queue lists of the batched path (allocated with the first batch)
    uint32_t* tile_mask = nullptr;        // [frames][mask_words]
    uint32_t* qlist = nullptr;            // [tiles * frames] live tile slots, queue after queue
    uint32_t* qscratch = nullptr;         // chunk_count | chunk_base | qcount
    int mask_words = 0, q_chunks_cap = 0;
    bool cull_on = true;
    bool frame_via_batch = false;
    int last_n_queues = 0;                // of the last batched launch (rto_ctx_queue_stats)
    int64_t last_slots = 0;
    int batch_fallback = 0;               // tuning / test hook, see rto_ctx_set_tuning
    int lean = 0;                         // rto_ctx_set_lean_outputs: 0 full, 1 lean, 2 lean + sparse
    std::vector<uint8_t> lean_slot;       // per frame slot: the level of its last writer (1: no aux planes, noisy.a = alpha; 2: and
                                          // nothing stored for the pixels of unmarked tiles)
    int test_wide_bits = 0;               // test hook "wide_bits": pretend a hit entry has this many bits for an entry of the two-level image
    bool cull_single = false;             // tuning "cull_single": the single-frame kernel culls too.  Off by default: a LONE frame waits
                                          // for its longest rays (marked tiles), and the two extra launches cost it 14 us (0.375 ->
                                          // 0.389 ms); with several frames in flight the skipped work is throughput (+6 %)
    int marks_n = 0, marks_slot0 = 0;     // frames whose tile marks the last launch left in tile_mask (0: none), their first slot
    float marks_bg = 0.f;                 // ... and the background brightness of that launch
    // per-kernel event timing of the batched path (off by default)
    bool kt_on = false;
    std::vector<hipEvent_t> kt_ev;  // kKtRing quadruples
    int kt_count = 0;               // launches recorded since the last read
    float* aux = nullptr;
    float* noisy = nullptr;
    float* image = nullptr;
    uint8_t* rgba8 = nullptr;
    rto::Pcg32 rng{};
    rto::PcgJumpEntry* jump = nullptr;
    uint64_t jump_inc = 0;
    bool jump_valid = false;
    int kernel = RTO_KERNEL_AUTO;
    int strip_rows = 1;
    int refill = 0;  // 0 = the default instantiation; 100 * waves/SIMD + idle-lane threshold picks an A/B one
    bool tile_order_on = true;
    bool stats_on = false;
    bool stats_marks = false;             // rto_ctx_enable_stats(2): count against the tile marks of the last batched launch
    unsigned long long* stats = nullptr;  // device, rto::kStatsWords counters
    // Timer (render_context.hpp:122-213)
    hipStream_t t_stream = nullptr;
    hipEvent_t t_start[3] = {nullptr, nullptr, nullptr}, t_stop[3] = {nullptr, nullptr, nullptr};
    bool t_used[3] = {false, false, false};
    float t_sum[3] = {0, 0, 0};
    int t_cnt = 0;
};

namespace {

void pcg_seed(rto::Pcg32& r, uint64_t initstate, uint64_t initseq) {  // pcg32.h:53-59
    auto next = [&]() { r.state = r.state * rto::kPcgMult + r.inc; };
    r.state = 0U;
    r.inc = (initseq << 1u) | 1u;
    next();
    r.state += initstate;
    next();
}

// (mult, plus) of the affine map "advance by delta" for increment `inc` (pcg32.h:145-166)
rto::PcgJumpEntry pcg_jump(uint64_t inc, uint64_t delta) {
    uint64_t cur_mult = rto::kPcgMult, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta /= 2;
    }
    return {acc_mult, acc_plus};
}

int ensure_jump_table(rto_ctx* c, hipStream_t stream) {
    if (c->jump_valid && c->jump_inc == c->rng.inc) return RTO_OK;
    std::vector<rto::PcgJumpEntry> tab(4 * 256);
    for (int ch = 0; ch < 4; ++ch)
        for (int j = 0; j < 256; ++j) tab[ch * 256 + j] = pcg_jump(c->rng.inc, (uint64_t)j << (8 * ch));
    if (!c->jump) HIP_TRY(hipMalloc((void**)&c->jump, tab.size() * sizeof(rto::PcgJumpEntry)));
    // synchronous copy from a stack-lifetime vector; happens once per `inc`
    HIP_TRY(hipMemcpy(c->jump, tab.data(), tab.size() * sizeof(rto::PcgJumpEntry), hipMemcpyHostToDevice));
    (void)stream;
    c->jump_inc = c->rng.inc;
    c->jump_valid = true;
    return RTO_OK;
}

// IEEE binary16 -> binary32 (exact), for the host-side look at a leaf's density
float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {  // subnormal: renormalise
            int e = -1;
            uint32_t m = man;
            do {
                ++e;
                m <<= 1;
            } while (!(m & 0x400u));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (m & 0x3ffu) << 13;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | man << 13;
    } else {
        bits = sign | (exp + 127 - 15) << 23 | man << 13;
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// The culling cells of a tree (TreeDev::occ_cells): cubes of size >= 2^-kOccLevel that together contain every leaf of positive
// density, as world-space bounding spheres.  `sigma(slot)` = the leaf's density.  Returns false when the node order does not
// allow the single top-down pass (a child stored before its parent: cannot happen after the breadth-first relayout).
template <class SigmaFn>
bool culling_cells(const int32_t* child, int64_t capacity, const float scale[3], const float offset[3], SigmaFn sigma,
                   std::vector<float>& out) {
    std::vector<uint8_t> lvl((size_t)capacity, 255), has((size_t)capacity, 0);
    std::vector<uint32_t> cx((size_t)capacity, 0), cy((size_t)capacity, 0), cz((size_t)capacity, 0);
    lvl[0] = 0;
    for (int64_t n = 0; n < capacity; ++n) {  // top-down: level and integer cell coordinates of every reachable node
        if (lvl[(size_t)n] == 255) continue;
        for (int s = 0; s < 8; ++s) {
            const int32_t c = child[n * 8 + s];
            if (c == 0) continue;
            const int64_t t = n + c;
            if (t <= n || t >= capacity || lvl[(size_t)n] >= 30) return false;
            lvl[(size_t)t] = (uint8_t)(lvl[(size_t)n] + 1);
            cx[(size_t)t] = cx[(size_t)n] * 2 + ((s >> 2) & 1);  // slot = x * 4 + y * 2 + z (n3tree_query.hpp:26-33)
            cy[(size_t)t] = cy[(size_t)n] * 2 + ((s >> 1) & 1);
            cz[(size_t)t] = cz[(size_t)n] * 2 + (s & 1);
        }
    }
    for (int64_t n = capacity - 1; n >= 0; --n) {  // bottom-up: does the subtree hold a leaf of positive density?
        if (lvl[(size_t)n] == 255) continue;
        uint8_t h = 0;
        for (int s = 0; s < 8 && !h; ++s) {
            const int32_t c = child[n * 8 + s];
            h = c ? has[(size_t)(n + c)] : (uint8_t)(sigma(n * 8 + s) > 0.f);
        }
        has[(size_t)n] = h;
    }
    out.clear();
    const float margin = 1e-4f;  // tree units: far above the float error of cen + t * dir (~1e-6), far below a cell
    for (int64_t n = 0; n < capacity; ++n) {
        if (lvl[(size_t)n] == 255 || !has[(size_t)n]) continue;
        const int ls = lvl[(size_t)n] + 1;  // the node's child slots are cubes of size 2^-ls
        if (ls > rto::kOccLevel) continue;  // inside a cube emitted above
        for (int s = 0; s < 8; ++s) {
            const int32_t c = child[n * 8 + s];
            const bool emit = c ? (ls == rto::kOccLevel && has[(size_t)(n + c)]) : sigma(n * 8 + s) > 0.f;
            if (!emit) continue;
            const double size = std::ldexp(1.0, -ls);
            const uint32_t q[3] = {cx[(size_t)n] * 2 + ((s >> 2) & 1), cy[(size_t)n] * 2 + ((s >> 1) & 1), cz[(size_t)n] * 2 + (s & 1)};
            double r2 = 0;
            for (int i = 0; i < 3; ++i) {
                const double ct = (q[i] + 0.5) * size;
                out.push_back((float)((ct - offset[i]) / scale[i]));
                const double hw = (0.5 * size + margin) / std::fabs((double)scale[i]);
                r2 += hw * hw;
            }
            out.push_back((float)(std::sqrt(r2) * 1.001));
        }
    }
    return true;
}

// Breadth-first node order of a tree: order[new] = old.  Children are visited in slot order, so after the
// renumbering the internal children of every node are consecutive, in slot order, and every level is stored
// in Morton order of its cells -- whatever order the file used (svox appends the children of
// whichever leaves a refinement step selected).  Nodes the root does not reach (spare capacity) keep their
// relative order behind the reachable ones.
std::vector<int64_t> bfs_order(const int32_t* child, int64_t capacity, int64_t N3) {
    std::vector<int64_t> order;
    order.reserve((size_t)capacity);
    std::vector<uint8_t> seen((size_t)capacity, 0);
    order.push_back(0);
    seen[0] = 1;
    for (size_t h = 0; h < order.size(); ++h) {
        const int64_t o = order[h];
        for (int64_t s = 0; s < N3; ++s) {
            const int32_t c = child[o * N3 + s];
            if (c == 0) continue;
            const int64_t t = o + c;  // in range: tree_max_depth validated every offset
            if (!seen[(size_t)t]) {
                seen[(size_t)t] = 1;
                order.push_back(t);
            }
        }
    }
    for (int64_t o = 0; o < capacity; ++o)
        if (!seen[(size_t)o]) order.push_back(o);
    return order;
}

// The same tree with its nodes stored in `order`: child offsets recomputed, per-slot arrays gathered.
struct Relaid {
    std::vector<int32_t> child;
    std::vector<uint16_t> data, q_map, q_sigma, q_retained;
};

void relay_tree(const std::vector<int64_t>& order, const int32_t* child, const uint16_t* data, int64_t capacity, int64_t N3,
                int data_dim, const rto::HostTree* quant, Relaid& out) {
    std::vector<int64_t> new_of_old((size_t)capacity);
    for (int64_t n = 0; n < capacity; ++n) new_of_old[(size_t)order[(size_t)n]] = n;
    out.child.resize((size_t)(capacity * N3));
    for (int64_t n = 0; n < capacity; ++n) {
        const int64_t o = order[(size_t)n];
        for (int64_t s = 0; s < N3; ++s) {
            const int32_t c = child[o * N3 + s];
            out.child[(size_t)(n * N3 + s)] = c ? (int32_t)(new_of_old[(size_t)(o + c)] - n) : 0;
        }
    }
    auto gather = [&](const uint16_t* src, size_t per_slot, std::vector<uint16_t>& dst) {  // [capacity*N3][per_slot]
        const size_t node_elems = (size_t)N3 * per_slot;
        for (int64_t n = 0; n < capacity; ++n)
            std::memcpy(dst.data() + (size_t)n * node_elems, src + (size_t)order[(size_t)n] * node_elems, node_elems * sizeof(uint16_t));
    };
    if (data) {
        out.data.resize((size_t)(capacity * N3) * (size_t)data_dim);
        gather(data, (size_t)data_dim, out.data);
    }
    if (quant) {
        const size_t ns = (size_t)(capacity * N3);
        const int nq = quant->n_basis - quant->n_retain, nr = quant->n_retain;
        out.q_sigma.resize(ns);
        gather(quant->q_sigma, 1, out.q_sigma);
        out.q_map.resize((size_t)nq * ns);
        for (int j = 0; j < nq; ++j) {
            std::vector<uint16_t> plane(ns);
            gather(quant->q_map + (size_t)j * ns, 1, plane);
            std::memcpy(out.q_map.data() + (size_t)j * ns, plane.data(), ns * sizeof(uint16_t));
        }
        out.q_retained.resize((size_t)nr * ns * 3);
        for (int j = 0; j < nr; ++j) {
            std::vector<uint16_t> plane(ns * 3);
            gather(quant->q_retained + (size_t)j * ns * 3, 3, plane);
            std::memcpy(out.q_retained.data() + (size_t)j * ns * 3, plane.data(), ns * 3 * sizeof(uint16_t));
        }
    }
}

// ---- two-level ("wide") traversal image for the batched traversal kernel (round 4) ----
// The persistent kernel visits one node per loop iteration; 0.65 of its 1.65 visits per march step are descents through
// internal nodes.  A wide node merges an octree node at level L = G + 2p (G = top-grid levels) with its eight children:
// 64 words, indexed by TWO bits per axis of the sample point, each holding what the two-level walk below that node ends in --
//   a leaf at level L (replicated into its 8 entries) or L + 1:  kLeafTag | level << 23 (kWideLevelShift) | sigma fp16   (the level rides in the
//                                                                 word because the entry no longer says which it was)
//   an internal node at level L + 2:                              the absolute index of ITS wide node
// so a walk costs one load per TWO levels.  Entry layout inside a wide node: (x2 << 4) | (y2 << 2) | z2 with x2 = the two
// bits (level L, level L + 1) of x: the eight entries below one child of the node share a 128-byte half.  Derived data:
// every (point -> leaf level, sigma, original leaf slot) answer equals the walk over child[] (tests/test_wide_image.py).
// worig[wide node] = its octree node, for translating a hit entry back to the leaf's slot in data[] / shrec[].
// ONE array holds the top grid and the wide nodes: entries [0, 8^G) are the grid cells (the "root node": G bits per axis),
// padded to whole nodes of 64; wide node k is node number grid_nodes + k of that array.  A walk is then uniform -- entry index
// = ((node << b | x bits) << b | y bits) << b | z bits with (node, b) = (0, G) at the grid and (node number, 2) below -- and an
// entry's index doubles as the hit index of its leaf.  gslot[grid cell] = the slot of a leaf cell above the grid levels.
struct WideImage {
    std::vector<uint32_t> widew, worig, gslot;
    uint32_t n_wide = 0, grid_nodes = 0;
};

template <class SigmaBits>
bool build_wide_image(const int32_t* child, int64_t capacity, int max_depth, int G, SigmaBits sigma_bits, WideImage& out) {
    // node ranges of the levels (the tree is stored breadth-first: a level's nodes are contiguous)
    std::vector<int64_t> start(1, 0), end(1, 1);
    for (int l = 0; l < 64; ++l) {
        int64_t hi = end[(size_t)l];
        for (int64_t n = start[(size_t)l]; n < end[(size_t)l]; ++n)
            for (int s = 0; s < 8; ++s) {
                const int32_t c = child[n * 8 + s];
                if (c != 0 && n + c + 1 > hi) hi = n + c + 1;
                if (c != 0 && n + c < end[(size_t)l]) return false;  // not breadth-first after all
            }
        if (hi == end[(size_t)l]) break;  // no children: the last level
        start.push_back(end[(size_t)l]);
        end.push_back(hi);
        if (hi > capacity) return false;
    }
    const int n_levels = (int)start.size();
    if (n_levels > 25 || G >= n_levels) return false;
    std::vector<int64_t> pair_base;  // first wide node of pair p
    int64_t n_wide = 0;
    for (int L = G; L < n_levels; L += 2) {
        pair_base.push_back(n_wide);
        n_wide += end[(size_t)L] - start[(size_t)L];
    }
    const int64_t grid_cells = int64_t(1) << (3 * G);
    const int64_t grid_nodes = (grid_cells + 63) / 64;
    if ((grid_nodes + n_wide) * 64 >= (int64_t(1) << rto::kGridSlotBits)) return false;
    out.n_wide = (uint32_t)n_wide;
    out.grid_nodes = (uint32_t)grid_nodes;
    out.widew.assign((size_t)(grid_nodes + n_wide) * 64, rto::kLeafTag);  // (padding reads as an empty leaf of level 0; never indexed)
    out.worig.assign((size_t)n_wide, 0u);
    auto leafw = [&](int level, int64_t slot) { return rto::kLeafTag | ((uint32_t)level << rto::kWideLevelShift) | (uint32_t)sigma_bits(slot); };
    auto entry = [](int a, int b) {  // child digits (x most significant) at level L and L + 1 -> position in the wide node
        const int x2 = ((a >> 2) & 1) << 1 | ((b >> 2) & 1), y2 = ((a >> 1) & 1) << 1 | ((b >> 1) & 1), z2 = (a & 1) << 1 | (b & 1);
        return x2 << 4 | y2 << 2 | z2;
    };
    for (size_t p = 0; p < pair_base.size(); ++p) {
        const int L = G + 2 * (int)p;
        for (int64_t N = start[(size_t)L]; N < end[(size_t)L]; ++N) {
            const int64_t wn = pair_base[p] + (N - start[(size_t)L]);
            out.worig[(size_t)wn] = (uint32_t)N;
            uint32_t* w = out.widew.data() + (size_t)(grid_nodes + wn) * 64;
            for (int a = 0; a < 8; ++a) {
                const int32_t c = child[N * 8 + a];
                if (c == 0) {
                    const uint32_t lw = leafw(L, N * 8 + a);
                    for (int b = 0; b < 8; ++b) w[entry(a, b)] = lw;
                    continue;
                }
                const int64_t C = N + c;
                for (int b = 0; b < 8; ++b) {
                    const int32_t c2 = child[C * 8 + b];
                    if (c2 == 0) {
                        w[entry(a, b)] = leafw(L + 1, C * 8 + b);
                    } else {
                        const int64_t D = C + c2;  // level L + 2: the first level of the next pair
                        if (L + 2 >= n_levels || D < start[(size_t)L + 2] || D >= end[(size_t)L + 2]) return false;
                        w[entry(a, b)] = (uint32_t)(grid_nodes + pair_base[p + 1] + (D - start[(size_t)L + 2]));
                    }
                }
            }
        }
    }
    // the top grid in the same terms (see build_topgrid_kernel): cell -> where its root path over levels 0..G-1 ends
    {
        const uint32_t mask = (1u << G) - 1u;
        out.gslot.assign((size_t)grid_cells, 0u);
        for (uint32_t key = 0; key < (uint32_t)grid_cells; ++key) {
            if (G == 0) {  // no grid levels: the one cell is the whole volume, below it the root's wide node
                out.widew[0] = (uint32_t)grid_nodes;
                break;
            }
            const uint32_t cx = key >> (2 * G), cy = (key >> G) & mask, cz = key & mask;
            int64_t node = 0, slot = 0;
            int32_t c = 0;
            int lvl = 0;
            for (;;) {
                const int sh = G - 1 - lvl;
                const uint32_t ci = (((cx >> sh) & 1u) << 2) | (((cy >> sh) & 1u) << 1) | ((cz >> sh) & 1u);
                slot = node * 8 + ci;
                c = child[slot];
                if (c == 0 || lvl == G - 1) break;
                node += c;
                ++lvl;
            }
            if (c == 0) {
                out.widew[key] = leafw(lvl, slot);
                out.gslot[key] = (uint32_t)slot;
            } else {  // internal at level G - 1: its child is a level-G node = a wide node of pair 0
                const int64_t D = node + c;
                if (D < start[(size_t)G] || D >= end[(size_t)G]) return false;
                out.widew[key] = (uint32_t)(grid_nodes + (D - start[(size_t)G]));
            }
        }
    }
    return true;
}

int upload_tree(const int32_t* child, const uint16_t* data, int64_t capacity, int N, int data_dim,
                const rto::DataFormat& fmt, const float scale[3], const float offset[3], int device,
                rto_tree** out, const rto::HostTree* quant = nullptr, int flags = 0) {
    if (!child || (!data && !quant) || capacity <= 0 || N < 1 || data_dim < 1 || !out)
        return set_err(RTO_E_INVALID, "rto_tree: null array or non-positive size");
    if (quant && (N != 2 || fmt.format != RTO_FMT_SH ||
                  !(fmt.basis_dim == 4 || fmt.basis_dim == 9 || fmt.basis_dim == 16 || fmt.basis_dim == 25) ||
                  quant->n_basis != fmt.basis_dim))
        return set_err(RTO_E_UNSUPPORTED, "direct rendering of quantised trees needs N == 2 and SH4/9/16/25 with matching codebooks");
    if (fmt.format == RTO_FMT_SH || fmt.format == RTO_FMT_SG || fmt.format == RTO_FMT_ASG) {
        if (fmt.basis_dim < 1 || data_dim != 3 * fmt.basis_dim + 1)
            return set_err(RTO_E_FORMAT, "rto_tree: data_dim " + std::to_string(data_dim) + " does not match format " +
                                             fmt.to_string());
        if (fmt.basis_dim > RTO_BASIS_MAX)
            return set_err(RTO_E_FORMAT, "rto_tree: basis_dim above " + std::to_string(RTO_BASIS_MAX));
    } else if (data_dim < 4) {
        return set_err(RTO_E_FORMAT, "rto_tree: RGBA trees need data_dim >= 4");
    }
    int max_depth = 0;
    try {
        max_depth = rto::tree_max_depth(child, capacity, N);
    } catch (const std::exception& e) {
        return set_err(RTO_E_FORMAT, e.what());
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_err(RTO_E_HIP, "no HIP device available (librto has no CPU fallback)");
    if (device < 0 || device >= ndev) return set_err(RTO_E_INVALID, "device index out of range");
    DeviceGuard guard(device);
    if (!guard.ok) return set_err(RTO_E_HIP, "hipSetDevice failed");

    auto t = new rto_tree();
    t->device = device;
    const int64_t N3 = (int64_t)N * N * N;
    const int64_t n_slots = capacity * N3;
    // Re-lay the tree breadth-first unless the file already is (derived layout: every query returns the same
    // leaf values, so no pixel can change -- tests/test_render_parity.py::test_node_order_never_changes_pixels)
    Relaid relaid;
    rto::HostTree quant_relaid;
    {
        const std::vector<int64_t> order = bfs_order(child, capacity, N3);
        bool identity = true;
        for (int64_t n = 0; n < capacity && identity; ++n) identity = order[(size_t)n] == n;
        if (!identity) {
            relay_tree(order, child, data, capacity, N3, data_dim, quant, relaid);
            child = relaid.child.data();
            if (data) data = relaid.data.data();
            if (quant) {
                quant_relaid = *quant;
                quant_relaid.q_map = relaid.q_map.data();
                quant_relaid.q_sigma = relaid.q_sigma.data();
                quant_relaid.q_retained = relaid.q_retained.empty() ? nullptr : relaid.q_retained.data();
                quant = &quant_relaid;
            }
        }
    }
    const size_t data_bytes = (size_t)n_slots * data_dim * sizeof(uint16_t);
    const size_t child_bytes = (size_t)n_slots * sizeof(int32_t);
    auto fail = [&](int code, const std::string& msg) {
        rto_tree_free(t);
        return set_err(code, msg);
    };
    size_t dev_bytes = child_bytes;
    if (hipMalloc(&t->d_child, child_bytes) != hipSuccess) return fail(RTO_E_HIP, "hipMalloc(tree.child) failed");
    if (hipMemcpy(t->d_child, child, child_bytes, hipMemcpyHostToDevice) != hipSuccess)
        return fail(RTO_E_HIP, "tree upload failed");
    int q_rec = 0;
    if (quant) {
        const int nq = quant->n_basis - quant->n_retain, nr = quant->n_retain;
        q_rec = (3 * nr + nq + 1) & ~1;  // u16 per slot record, dword aligned
        const size_t map_b = (size_t)nq * n_slots * 2, ret_b = (size_t)nr * n_slots * 3 * 2, sig_b = (size_t)n_slots * 2;
        const size_t rec_b = (size_t)n_slots * q_rec * 2, col_b = (size_t)nq * 65536 * 8;
        // codebooks: {r,g,b} fp16 -> {r,g,b,0} so that one 8 B load fetches an entry
        std::vector<uint16_t> cb((size_t)nq * 65536 * 4, 0);
        for (size_t i = 0; i < (size_t)nq * 65536; ++i)
            for (int c = 0; c < 3; ++c) cb[i * 4 + c] = quant->q_colors[i * 3 + c];
        void *tmp_map = nullptr, *tmp_ret = nullptr;
        auto up = [&](void** dst, const void* src, size_t bytes) {
            if (bytes == 0) return true;
            return hipMalloc(dst, bytes) == hipSuccess && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
        };
        // file layout up, re-laid slot-major on the device, staging freed (peak = 2x the quantised arrays)
        bool ok = up(&tmp_map, quant->q_map, map_b) && up(&tmp_ret, quant->q_retained, ret_b) &&
                  up(&t->d_qcolors, cb.data(), col_b) && up(&t->d_qsigma, quant->q_sigma, sig_b) &&
                  hipMalloc(&t->d_qrec, rec_b) == hipSuccess &&
                  rto::launch_pack_quant((const uint16_t*)tmp_map, (const uint16_t*)tmp_ret, n_slots, nr, nq, q_rec,
                                         (uint16_t*)t->d_qrec, nullptr) == hipSuccess &&
                  hipDeviceSynchronize() == hipSuccess;
        if (tmp_map) (void)hipFree(tmp_map);
        if (tmp_ret) (void)hipFree(tmp_ret);
        if (!ok) return fail(RTO_E_HIP, "quantised tree upload failed");
        t->quant = true;
        dev_bytes += rec_b + col_b + sig_b;
    } else {
        // +16 B: shade_leaf_packed reads whole dwords around a record and may touch up to 4 B past it
        if (hipMalloc(&t->d_data, data_bytes + 16) != hipSuccess) return fail(RTO_E_HIP, "hipMalloc(tree.data) failed");
        if (hipMemcpy(t->d_data, data, data_bytes, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset((char*)t->d_data + data_bytes, 0, 16) != hipSuccess)
            return fail(RTO_E_HIP, "tree upload failed");
        dev_bytes += data_bytes + 16;
    }
    // traversal image for the fast kernel: N == 2, depth within the 24 fixed-point bits, slot index within the 29
    // bits of a top-grid entry (the hit-list budget depends on the SPP and is checked per launch: slots_fit_spp)
    if (N == 2 && max_depth <= 24 && n_slots < (int64_t(1) << rto::kGridSlotBits)) {
        int* d_bad = nullptr;
        if (hipMalloc(&t->d_nodew, (size_t)n_slots * 4) != hipSuccess || hipMalloc((void**)&d_bad, 4) != hipSuccess ||
            hipMemset(d_bad, 0, 4) != hipSuccess)
            return fail(RTO_E_HIP, "hipMalloc(nodew) failed");
        // sigma source: the last value of each dense record, or the quantised set's own sigma array
        hipError_t e = quant ? rto::launch_build_nodew((const int32_t*)t->d_child, (const uint16_t*)t->d_qsigma, n_slots, 1,
                                                       (uint32_t*)t->d_nodew, d_bad, nullptr)
                             : rto::launch_build_nodew((const int32_t*)t->d_child, (const uint16_t*)t->d_data, n_slots,
                                                       data_dim, (uint32_t*)t->d_nodew, d_bad, nullptr);
        int bad = 0;
        if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
        (void)hipFree(d_bad);
        if (e != hipSuccess) return fail(RTO_E_HIP, std::string("build_nodew failed: ") + hipGetErrorString(e));
        t->fast_ok = !bad;
        if (bad) {
            (void)hipFree(t->d_nodew);
            t->d_nodew = nullptr;
        } else {
            dev_bytes += (size_t)n_slots * 4;
        }
    }

    if (t->fast_ok && !(flags & RTO_TREE_NO_CULLING)) {  // culling cells for the batched path (see TreeDev::occ_cells)
        std::vector<float> cells;
        bool ok;
        if (quant) {
            const uint16_t* qs = quant->q_sigma;
            ok = culling_cells(child, capacity, scale, offset, [&](int64_t sl) { return half_to_float(qs[sl]); }, cells);
        } else {
            const size_t dd = (size_t)data_dim;
            ok = culling_cells(child, capacity, scale, offset, [&](int64_t sl) { return half_to_float(data[(size_t)sl * dd + dd - 1]); }, cells);
        }
        if (ok) {
            const size_t bytes = cells.empty() ? 16 : cells.size() * sizeof(float);
            if (hipMalloc(&t->d_occ, bytes) != hipSuccess ||
                (!cells.empty() && hipMemcpy(t->d_occ, cells.data(), cells.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess))
                return fail(RTO_E_HIP, "hipMalloc(culling cells) failed");
            t->dev.occ_cells = (const float4*)t->d_occ;
            t->dev.n_occ_cells = (int)(cells.size() / 4);
            dev_bytes += bytes;
        }
    }

    int top_levels = 0;
    if (t->fast_ok && max_depth >= 3) {
        // shortcut grid over the top levels: 2^(3G) x 8 B (2 MB at G = 6: L2-resident)
        // (6 levels: measured again in round 3 with 7 waves per SIMD -- 5 levels 7.1-7.25 ms per 100 frames, 6: 6.88-7.0,
        //  7 (16 MB): 6.85-6.96, 8 (134 MB): 6.93-6.99)
        top_levels = max_depth - 1 < 6 ? max_depth - 1 : 6;
        if (const char* ev = getenv("RTO_TOP_LEVELS")) {  // (A/B hook: levels the top grid covers, 3..8; same pixels for every value)
            const int g = atoi(ev);
            if (g >= 3 && g <= 8 && g <= max_depth - 1) top_levels = g;
        }
        const size_t gbytes = (size_t)8 << (3 * top_levels);
        if (hipMalloc(&t->d_topgrid, gbytes) != hipSuccess) return fail(RTO_E_HIP, "hipMalloc(topgrid) failed");
        hipError_t e = rto::launch_build_topgrid((const uint32_t*)t->d_nodew, top_levels, (uint2*)t->d_topgrid, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) return fail(RTO_E_HIP, std::string("build_topgrid failed: ") + hipGetErrorString(e));
        dev_bytes += gbytes;
    }

    if (t->fast_ok && !getenv("RTO_NO_WIDE")) {  // the two-level traversal image of the batched kernel (build_wide_image)
        WideImage wi;
        bool ok;
        if (quant) {
            const uint16_t* qs = quant->q_sigma;
            ok = build_wide_image(child, capacity, max_depth, top_levels, [&](int64_t sl) { return qs[sl]; }, wi);
        } else {
            const size_t dd = (size_t)data_dim;
            ok = build_wide_image(child, capacity, max_depth, top_levels, [&](int64_t sl) { return data[(size_t)sl * dd + dd - 1]; }, wi);
        }
        if (ok) {
            const size_t wb = wi.widew.size() * 4, ob = wi.worig.size() * 4, gb = wi.gslot.size() * 4;
            const bool up = hipMalloc(&t->d_widew, wb) == hipSuccess && hipMalloc(&t->d_worig, ob) == hipSuccess &&
                            hipMalloc(&t->d_grid2, gb) == hipSuccess &&
                            hipMemcpy(t->d_widew, wi.widew.data(), wb, hipMemcpyHostToDevice) == hipSuccess &&
                            hipMemcpy(t->d_worig, wi.worig.data(), ob, hipMemcpyHostToDevice) == hipSuccess &&
                            hipMemcpy(t->d_grid2, wi.gslot.data(), gb, hipMemcpyHostToDevice) == hipSuccess;
            if (up) {
                t->dev.widew = (const uint32_t*)t->d_widew;
                t->dev.worig = (const uint32_t*)t->d_worig;
                t->dev.wgslot = (const uint32_t*)t->d_grid2;
                t->dev.wide_entries = wi.n_wide * 64u;
                t->dev.wide_grid_nodes = wi.grid_nodes;
                dev_bytes += wb + ob + gb;
            } else {  // not enough memory: the kernel walks the one-level image
                (void)hipGetLastError();
                for (void** q : {&t->d_widew, &t->d_worig, &t->d_grid2}) {
                    if (*q) (void)hipFree(*q);
                    *q = nullptr;
                }
            }
        }
    }

    if (t->fast_ok && !quant && fmt.format == RTO_FMT_SH && (fmt.basis_dim == 9 || fmt.basis_dim == 16) &&
        !(flags & RTO_TREE_COMPACT)) {
        // aligned copy of the SH coefficients for the shading kernels (+ 64 / 128 B per slot)
        const int rec = 3 * fmt.basis_dim * 2 <= 64 ? 32 : 64;  // = shrec_halves()
        int64_t n_rec = n_slots;
        if (flags & RTO_TREE_COMPACT_RECORDS) {
            // records only for the leaf slots a ray can hit -- density > 0 (a hit needs sigma > sigma_thresh >= 0,
            // rt_core.cuh:252; launches with a negative threshold are refused for such a tree) -- in slot order
            std::vector<uint32_t> idx((size_t)n_slots, rto::kNoRecord);
            uint32_t n = 0;
            for (int64_t sl = 0; sl < n_slots; ++sl) {
                if (child[sl] != 0) continue;
                const float sg = half_to_float(data[(size_t)sl * data_dim + data_dim - 1]);
                if (sg > 0.f) idx[(size_t)sl] = n++;
            }
            n_rec = n > 0 ? n : 1;
            if (hipMalloc(&t->d_recidx, (size_t)n_slots * 4) != hipSuccess ||
                hipMemcpy(t->d_recidx, idx.data(), (size_t)n_slots * 4, hipMemcpyHostToDevice) != hipSuccess)
                return fail(RTO_E_HIP, "hipMalloc(record index) failed");
            dev_bytes += (size_t)n_slots * 4;
        }
        // Round 5: with a two-level traversal image the records follow ITS entries (TreeDev::rec_by_entry), so that a hit entry
        // names its record and nothing translates between traversal and shading.  Not with compact records (their index is
        // per slot), not when the entries would not fit the hit-entry budget of every SPP (2^26 at SPP 32: such a tree must
        // stay able to fall back to the one-level walk, whose hits name slots), not with RTO_TREE_SLOT_RECORDS (A/B, tests).
        const int64_t n_entries = t->dev.widew ? (int64_t)t->dev.wide_entries + (int64_t)t->dev.wide_grid_nodes * 64 : 0;
        // (the gate is fast_path_for_spp's own test at the largest supported SPP: a by-entry tree has no one-level fallback)
        bool by_entry = n_entries > 0 && !(flags & RTO_TREE_COMPACT_RECORDS) && slots_fit_spp(n_entries, kMaxSpp) && !getenv("RTO_TREE_SLOT_RECORDS");
        const int64_t n_rec_slots = n_rec;
        if (by_entry) n_rec = n_entries;
        size_t rb = (size_t)n_rec * rec * 2;
        bool got = hipMalloc(&t->d_shrec, rb) == hipSuccess && !(by_entry && getenv("RTO_TEST_FAIL_ENTRY_RECORDS"));
        if (!got && by_entry) {  //  the entry-ordered copy (up to ~8x the slots) does not fit: the slot-ordered one may
            (void)hipGetLastError();
            if (t->d_shrec) (void)hipFree(t->d_shrec);
            t->d_shrec = nullptr;
            by_entry = false;
            n_rec = n_rec_slots;
            rb = (size_t)n_rec * rec * 2;
            got = hipMalloc(&t->d_shrec, rb) == hipSuccess;
        }
        if (got) {
            hipError_t e;
            if (by_entry) {
                rto::TreeDev td = t->dev;  // (the look-up tables of the two-level image are set; the rest of it as far as the kernel reads it)
                td.nodew = (const uint32_t*)t->d_nodew;
                td.data_dim = data_dim;
                td.top_levels = top_levels;
                e = rto::launch_build_shrec_wide(td, (const uint16_t*)t->d_data, n_entries, rec, (uint16_t*)t->d_shrec, nullptr);
                t->dev.rec_by_entry = 1;
            } else
                e = rto::launch_build_shrec((const uint16_t*)t->d_data, n_slots, data_dim, rec, (const uint32_t*)t->d_recidx,
                                            (uint16_t*)t->d_shrec, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess) return fail(RTO_E_HIP, std::string("build_shrec failed: ") + hipGetErrorString(e));
            dev_bytes += rb;
            t->shrec_halves = rec;
        } else {  // not enough memory for the copy: shade from data[]
            (void)hipGetLastError();
            t->d_shrec = nullptr;
            if (t->d_recidx) {
                (void)hipFree(t->d_recidx);
                t->d_recidx = nullptr;
                dev_bytes -= (size_t)n_slots * 4;
            }
        }
    }

    if (t->d_shrec && !(flags & RTO_TREE_KEEP_REFERENCE)) {
        // the fast / batched kernels never read child[] / data[] of such a tree: release them (rebuilt on demand)
        (void)hipFree(t->d_data);
        (void)hipFree(t->d_child);
        t->d_data = t->d_child = nullptr;
        t->reference_dropped = true;
        dev_bytes -= data_bytes + 16 + child_bytes;
    }

    rto::TreeDev& d = t->dev;
    if (t->d_qsigma) {  // sigma now lives in the traversal image
        (void)hipFree(t->d_qsigma);
        t->d_qsigma = nullptr;
        dev_bytes -= (size_t)n_slots * 2;
    }
    if (quant && !t->fast_ok) return fail(RTO_E_UNSUPPORTED, "direct rendering of quantised trees needs the N == 2 traversal image");
    d.topgrid = (const uint2*)t->d_topgrid;
    d.top_levels = top_levels;
    d.shrec = (const uint16_t*)t->d_shrec;
    d.recidx = (const uint32_t*)t->d_recidx;
    d.qrec = (const uint16_t*)t->d_qrec;
    d.qcolors = (const uint2*)t->d_qcolors;
    d.q_retain = quant ? quant->n_retain : 0;
    d.q_rec = q_rec;
    d.data = (const uint16_t*)t->d_data;
    d.child = (const int32_t*)t->d_child;
    d.nodew = (const uint32_t*)t->d_nodew;
    for (int i = 0; i < 3; ++i) {
        d.offset[i] = offset[i];
        d.scale[i] = scale[i];
    }
    d.N = N;
    d.N3 = (int)N3;
    d.data_dim = data_dim;
    d.format = fmt.format;
    d.basis_dim = fmt.basis_dim;
    d.ndc_width = -1.f;  // data_spec.hpp:49
    d.ndc_height = 0.f;
    d.ndc_focal = 0.f;
    d.max_depth = max_depth;

    rto_tree_info& inf = t->info;
    inf.capacity = capacity;
    inf.N = N;
    inf.data_dim = data_dim;
    inf.format = fmt.format;
    inf.basis_dim = fmt.basis_dim;
    for (int i = 0; i < 3; ++i) {
        inf.scale[i] = scale[i];
        inf.offset[i] = offset[i];
    }
    inf.use_ndc = 0;
    inf.ndc_width = inf.ndc_height = inf.ndc_focal = 0.f;
    inf.max_depth = max_depth;
    inf.device_bytes = (int64_t)dev_bytes;
    inf.wide_nodes = d.widew ? (int64_t)(d.wide_entries / 64u) : 0;
    *out = t;
    return RTO_OK;
}

// child[] / data[] of a tree that dropped them at upload, back on the device for the generic kernel (same leaf values:
// tests/test_render_parity.py::test_generic_kernel_on_a_tree_without_reference_arrays)
bool tree_fits_spp(const rto_tree* tree, int spp) { return fast_path_for_spp(tree, spp, 0, nullptr) != 0; }

int fast_path_for_spp(const rto_tree* tree, int spp, int wide_bits, rto::TreeDev* td) {
    if (td) *td = tree->dev;
    if (!tree->fast_ok) return 0;
    const int64_t n_slots = tree->info.capacity * tree->dev.N3;
    const int64_t wide = tree->dev.widew ? (int64_t)tree->dev.wide_entries + (int64_t)tree->dev.wide_grid_nodes * 64 : 0;
    const bool wide_fits = wide_bits > 0 ? wide < (int64_t(1) << wide_bits) : slots_fit_spp(wide, spp);
    if (tree->dev.rec_by_entry) return wide_fits ? 1 : 0;  // (its records are indexed by the entries: no one-level fallback; the
                                                           //  upload only chooses that layout when the entries fit every SPP)
    if (!slots_fit_spp(n_slots, spp)) return 0;
    if (wide_fits) return 1;
    // the entry count of the two-level image can be ~8x the slot count of the deepest level (odd number of levels
    // below the grid), so it can exceed the budget where the slots do not -- such a launch walks the one-level image
    // (nodew + topgrid, the WIDE = false instantiations) instead of dropping to the generic kernel
    if (td) {
        td->widew = nullptr;
        td->wgslot = nullptr;
        td->worig = nullptr;
        td->wide_entries = 0;
        td->wide_grid_nodes = 0;
    }
    return 2;
}

int ensure_reference_arrays(const rto_tree* tree) {
    if (!tree->reference_dropped) return RTO_OK;
    rto_tree* t = const_cast<rto_tree*>(tree);  // derived, lazily materialised state of a logically const tree
    std::lock_guard<std::mutex> lock(t->rebuild_mutex);
    if (!t->reference_dropped) return RTO_OK;
    DeviceGuard guard(t->device);
    if (!guard.ok) return set_err(RTO_E_HIP, "hipSetDevice failed");
    const int64_t n_slots = t->info.capacity * t->dev.N3;
    const size_t data_bytes = (size_t)n_slots * t->dev.data_dim * sizeof(uint16_t), child_bytes = (size_t)n_slots * sizeof(int32_t);
    void *d_data = nullptr, *d_child = nullptr;
    if (hipMalloc(&d_data, data_bytes + 16) != hipSuccess || hipMalloc(&d_child, child_bytes) != hipSuccess) {
        if (d_data) (void)hipFree(d_data);
        (void)hipGetLastError();
        return set_err(RTO_E_HIP, "the generic kernel needs the tree's child[] / data[] arrays back on the device: hipMalloc failed");
    }
    hipError_t e = hipMemset((char*)d_data + data_bytes, 0, 16);
    if (e == hipSuccess)
        e = rto::launch_rebuild_reference(t->dev.rec_by_entry ? nullptr : (const uint16_t*)t->d_shrec, (const uint32_t*)t->d_nodew,
                                          (const uint32_t*)t->d_recidx, n_slots, t->dev.data_dim, t->shrec_halves, (uint16_t*)d_data,
                                          (int32_t*)d_child, nullptr);
    if (e == hipSuccess && t->dev.rec_by_entry)  // entry-ordered records: the coefficients in a second pass, entry by entry
        e = rto::launch_rebuild_reference_wide(t->dev, (int64_t)t->dev.wide_entries + (int64_t)t->dev.wide_grid_nodes * 64, t->shrec_halves,
                                               (uint16_t*)d_data, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(d_data);
        (void)hipFree(d_child);
        return set_err(RTO_E_HIP, std::string("rebuilding child[] / data[] failed: ") + hipGetErrorString(e));
    }
    t->d_data = d_data;
    t->d_child = d_child;
    t->dev.data = (const uint16_t*)d_data;
    t->dev.child = (const int32_t*)d_child;
    t->info.device_bytes += (int64_t)(data_bytes + 16 + child_bytes);
    t->reference_dropped = false;
    return RTO_OK;
}

// RenderOptions -> the fields the kernels read.  rodrigues (volrend.cu:58-73): angle, axis, cos and sin
// depend on the options only, so they are evaluated here once per launch (float arithmetic, libm cosf /
// sinf: the same calls the CPU oracle makes) instead of once per pixel.
rto::OptDev make_opt_dev(const rto_options* o) {
    rto::OptDev od;
    od.step_size = o->step_size;
    od.sigma_thresh = o->sigma_thresh;
    od.background_brightness = o->background_brightness;
    std::memcpy(od.render_bbox, o->render_bbox, sizeof(od.render_bbox));
    od.basis_minmax[0] = o->basis_minmax[0];
    od.basis_minmax[1] = o->basis_minmax[1];
    const float* a = o->rot_dirs;
    const float angle = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);  // _norm common.cuh:16-20
    od.rot_on = !(angle < 1e-6);
    od.rot_k[0] = od.rot_k[1] = od.rot_k[2] = 0.f;
    od.rot_cos = 1.f;
    od.rot_sin = 0.f;
    od.rot_omc = 0.0;
    if (od.rot_on) {
        for (int i = 0; i < 3; ++i) od.rot_k[i] = a[i] / angle;
        od.rot_cos = cosf(angle);
        od.rot_sin = sinf(angle);
        od.rot_omc = 1.0 - od.rot_cos;
    }
    return od;
}

int options_from_value(const rto::json::Value& j, rto_options* o) {
    rto_options r;
    rto_options_default(&r);
    try {
        // NLOHMANN_DEFINE_TYPE_INTRUSIVE (render_options.hpp:61-77): every listed key is required
        r.step_size = (float)j.at("step_size").as_number();
        r.sigma_thresh = (float)j.at("sigma_thresh").as_number();
        r.stop_thresh = (float)j.at("stop_thresh").as_number();
        r.background_brightness = (float)j.at("background_brightness").as_number();
        r.show_grid = j.at("show_grid").as_bool();
        r.grid_max_depth = (int)j.at("grid_max_depth").as_number();
        r.enable_probe = j.at("enable_probe").as_bool();
        const auto& p = j.at("probe");
        if (p.size() != 3) throw std::runtime_error("json: 'probe' must have 3 elements");
        for (int i = 0; i < 3; ++i) r.probe[i] = (float)p.at(i).as_number();
        r.probe_disp_size = (int)j.at("probe_disp_size").as_number();
        r.denoise = j.at("denoise").as_bool();
        r.spp = (int)j.at("spp").as_number();
    } catch (const std::exception& e) {
        return set_err(RTO_E_FORMAT, std::string("render options: ") + e.what());
    }
    *o = r;
    return RTO_OK;
}

}  // namespace

// error hook for the other translation units of the library (not part of the public ABI)
extern "C" int rto_set_error_(int code, const char* msg) { return set_err(code, msg ? msg : ""); }

extern "C" {

const char* rto_version(void) { return "rt-octree_amd 0.1 (gfx950)"; }
const char* rto_last_error(void) { return g_err.c_str(); }

int rto_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return set_err(RTO_E_HIP, "hipGetDeviceCount failed");
    return n;
}

void rto_options_default(rto_options* o) {  // render_options.hpp:15-58
    if (!o) return;
    o->step_size = 1e-4f;
    o->sigma_thresh = 1e-2f;
    o->stop_thresh = 1e-2f;
    o->background_brightness = 1.f;
    const float bb[6] = {0.f, 0.f, 0.f, 1.f, 1.f, 1.f};
    std::memcpy(o->render_bbox, bb, sizeof(bb));
    o->basis_minmax[0] = 0;
    o->basis_minmax[1] = RTO_BASIS_MAX - 1;
    o->rot_dirs[0] = o->rot_dirs[1] = o->rot_dirs[2] = 0.f;
    o->show_grid = 0;
    o->grid_max_depth = 4;
    o->render_depth = 0;
    o->enable_probe = 0;
    o->probe[0] = 0.f;
    o->probe[1] = 0.f;
    o->probe[2] = 1.f;
    o->probe_disp_size = 100;
    o->denoise = 1;
    o->spp = 1;
}

int rto_options_from_json(const char* text, rto_options* o) {
    if (!text || !o) return set_err(RTO_E_INVALID, "rto_options_from_json: null argument");
    rto::json::ValuePtr v;
    try {
        v = rto::json::parse(text);
    } catch (const std::exception& e) {
        return set_err(RTO_E_FORMAT, e.what());
    }
    return options_from_value(*v, o);
}

int rto_options_from_json_file(const char* path, rto_options* o) {
    if (!path || !o) return set_err(RTO_E_INVALID, "rto_options_from_json_file: null argument");
    std::ifstream f(path);
    if (!f) return set_err(RTO_E_IO, std::string("cannot open options file '") + path + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    return rto_options_from_json(ss.str().c_str(), o);
}

int rto_tree_load_npz(const char* path, int device, rto_tree** out) { return rto_tree_load_npz_ex(path, device, 0, out); }

int rto_tree_load_npz_ex(const char* path, int device, int flags, rto_tree** out) {
    if (!path || !out) return set_err(RTO_E_INVALID, "rto_tree_load_npz: null argument");
    rto::HostTree h;
    try {
        if (!h.open(path, (flags & RTO_TREE_QUANT_DIRECT) != 0))
            return set_err(RTO_E_IO, std::string("file does not exist: ") + path);
    } catch (const std::exception& e) {
        return set_err(RTO_E_FORMAT, e.what());
    }
    std::fprintf(stdout, "INFO: Scale %f %f %f\n", h.scale[0], h.scale[1], h.scale[2]);  // n3tree.cpp:264
    int rc = upload_tree(h.child, h.data, h.capacity, h.N, h.data_dim, h.data_format, h.scale, h.offset, device, out,
                         h.quantized ? &h : nullptr, flags);
    if (rc != RTO_OK) return rc;
    if (h.use_ndc) rto_tree_set_ndc(*out, h.ndc_width, h.ndc_height, h.ndc_focal);
    return RTO_OK;
}

int rto_tree_probe_npz(const char* path, char* json_out, size_t cap) {
    if (!path || !json_out || cap == 0) return set_err(RTO_E_INVALID, "rto_tree_probe_npz: null argument");
    rto::HostTree h;
    try {
        if (!h.open(path)) return set_err(RTO_E_IO, std::string("file does not exist: ") + path);
    } catch (const std::exception& e) {
        return set_err(RTO_E_FORMAT, e.what());
    }
    auto fnv = [](const void* p, size_t n) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        uint64_t x = 1469598103934665603ULL;
        for (size_t i = 0; i < n; ++i) {
            x ^= b[i];
            x *= 1099511628211ULL;
        }
        return x;
    };
    const size_t n_slots = (size_t)h.capacity * h.N * h.N * h.N;
    int max_depth = 0;
    try {
        max_depth = rto::tree_max_depth(h.child, h.capacity, h.N);
    } catch (const std::exception& e) {
        return set_err(RTO_E_FORMAT, e.what());
    }
    char buf[1024];
    const int n = std::snprintf(
        buf, sizeof(buf),
        "{\"capacity\": %lld, \"N\": %d, \"data_dim\": %d, \"data_format\": \"%s\", \"basis_dim\": %d, "
        "\"scale\": [%.9g, %.9g, %.9g], \"offset\": [%.9g, %.9g, %.9g], \"use_ndc\": %d, \"max_depth\": %d, "
        "\"quantized\": %d, \"child_fnv1a64\": \"%016llx\", \"data_fnv1a64\": \"%016llx\"}",
        (long long)h.capacity, h.N, h.data_dim, h.data_format.to_string().c_str(), h.data_format.basis_dim,
        h.scale[0], h.scale[1], h.scale[2], h.offset[0], h.offset[1], h.offset[2], (int)h.use_ndc, max_depth,
        (int)!h.decoded.empty(), (unsigned long long)fnv(h.child, n_slots * 4),
        (unsigned long long)fnv(h.data, n_slots * (size_t)h.data_dim * 2));
    if (n < 0 || (size_t)n + 1 > cap) return set_err(RTO_E_INVALID, "rto_tree_probe_npz: output buffer too small");
    std::memcpy(json_out, buf, (size_t)n + 1);
    return RTO_OK;
}

int rto_tree_from_arrays(const int32_t* child, const uint16_t* data, int64_t capacity, int N, int data_dim,
                         const char* data_format, const float scale[3], const float offset[3], int device,
                         rto_tree** out) {
    return rto_tree_from_arrays_ex(child, data, capacity, N, data_dim, data_format, scale, offset, device, 0, out);
}

int rto_tree_from_arrays_ex(const int32_t* child, const uint16_t* data, int64_t capacity, int N, int data_dim,
                            const char* data_format, const float scale[3], const float offset[3], int device, int flags,
                            rto_tree** out) {
    if (!scale || !offset) return set_err(RTO_E_INVALID, "rto_tree_from_arrays: null scale/offset");
    rto::DataFormat fmt;
    if (data_format && data_format[0]) {
        fmt.parse(data_format);
    } else if (data_dim == 4) {  // n3tree.cpp:241-254 legacy autodetect
        fmt.format = RTO_FMT_RGBA;
        fmt.basis_dim = -1;
    } else {
        fmt.format = RTO_FMT_SH;
        fmt.basis_dim = (data_dim - 1) / 3;
    }
    return upload_tree(child, data, capacity, N, data_dim, fmt, scale, offset, device, out, nullptr, flags & ~RTO_TREE_QUANT_DIRECT);
}

int rto_tree_set_ndc(rto_tree* t, float w, float h, float focal) {
    if (!t) return set_err(RTO_E_INVALID, "rto_tree_set_ndc: null tree");
    t->info.use_ndc = w > 0;
    t->info.ndc_width = w;
    t->info.ndc_height = h;
    t->info.ndc_focal = focal;
    t->dev.ndc_width = w > 0 ? w : -1.f;
    t->dev.ndc_height = h;
    t->dev.ndc_focal = focal;
    return RTO_OK;
}

int rto_tree_get_info(const rto_tree* t, rto_tree_info* info) {
    if (!t || !info) return set_err(RTO_E_INVALID, "rto_tree_get_info: null argument");
    *info = t->info;
    return RTO_OK;
}

void rto_tree_free(rto_tree* t) {
    if (!t) return;
    DeviceGuard guard(t->device);
    if (t->d_data) (void)hipFree(t->d_data);
    if (t->d_child) (void)hipFree(t->d_child);
    if (t->d_nodew) (void)hipFree(t->d_nodew);
    if (t->d_topgrid) (void)hipFree(t->d_topgrid);
    for (void* q : {t->d_widew, t->d_worig, t->d_grid2})
        if (q) (void)hipFree(q);
    if (t->d_shrec) (void)hipFree(t->d_shrec);
    if (t->d_recidx) (void)hipFree(t->d_recidx);
    if (t->d_occ) (void)hipFree(t->d_occ);
    for (void* p : {t->d_qrec, t->d_qcolors, t->d_qsigma})
        if (p) (void)hipFree(p);
    delete t;
}

int rto_ctx_create(int width, int height, int device, rto_ctx** out) {
    return rto_ctx_create_batch(width, height, 1, device, out);
}


// Queue orders of the 8x8 ray tiles (persistent kernel).
//  * tile_order: rings around the image centre, innermost first, each ring walked by angle -- the
//    frame's long rays (the object) start early, the queue ends on cheap border tiles, consecutive
//    tiles stay neighbours;
//  * wedge_order: the tiles of the 8 XCD ray queues.  Rounds 3-5: the image cut into 8 angular wedges around the centre, blocks of
//    tile_block x tile_block tiles centre-out, Morton order inside a block.  Round 6 (queue_bands > 0, the default): BANDS of
//    queue_bands tile rows, band j -> queue j % 8, centre bands first, a band's tiles centre-out.  Every queue still gets its share
//    of the expensive centre and of the cheap border (the bands interleave), but what its XCD's L2 has to hold is a few horizontal
//    slabs of the tree: a camera that orbits the scene's vertical axis keeps a leaf in its rows from frame to frame, while an
//    angular wedge of the image sees the whole scene turn past it over the batch.
static int build_tile_tables(rto_ctx* c) {
    const int tx8 = (c->width + 7) / 8, ty8 = (c->height + 7) / 8;
    const int B = c->tile_block < 1 ? 1 : c->tile_block;
    struct Keyed { int wedge; double ring, ang; uint32_t morton; uint32_t code; };
    std::vector<Keyed> keyed;
    keyed.reserve((size_t)tx8 * ty8);
    const double cx = 0.5 * (tx8 - 1), cy = 0.5 * (ty8 - 1);
    const double pi = 3.14159265358979323846;
    auto polar = [&](double x, double y, double& ring, double& ang) {
        const double dx = x - cx, dy = y - cy;
        ring = std::floor(std::fmax(std::fabs(dx), std::fabs(dy)) + 0.5);
        ang = std::atan2(dy, dx) + pi;  // [0, 2pi]
    };
    auto spread = [](uint32_t v) {  // interleave helper for up to 8 bits
        uint32_t r = 0;
        for (int i = 0; i < 8; ++i) r |= ((v >> i) & 1u) << (2 * i);
        return r;
    };
    std::vector<std::pair<double, uint32_t>> plain;
    plain.reserve((size_t)tx8 * ty8);
    for (int ty = 0; ty < ty8; ++ty)
        for (int tx = 0; tx < tx8; ++tx) {
            const uint32_t code = ((uint32_t)ty << 16) | (uint32_t)tx;
            double ring, ang;
            polar(tx, ty, ring, ang);
            plain.emplace_back(ring * 16.0 + ang, code);
            // wedge / order key of the block the tile belongs to (block centre, in tile units)
            const int bx = tx / B, by = ty / B;
            double bring, bang;
            polar(bx * B + 0.5 * (B - 1), by * B + 0.5 * (B - 1), bring, bang);
            int wedge = (int)(bang / (2.0 * pi) * rto::kMaxQueues);
            if (wedge >= rto::kMaxQueues) wedge = rto::kMaxQueues - 1;
            if (c->queue_bands > 0) {
                // BANDS (round 6): band j of queue_bands tile rows -> queue j % 8.  A camera that orbits the scene's vertical axis keeps
                // a leaf in its rows from frame to frame, so a queue's rays -- the same tiles of all frames of the batch in turn --
                // stay inside a few horizontal slabs of the tree, which is what that XCD's L2 then has to hold; an angular wedge of
                // the image sees the whole scene turn past it.  Centre bands first, a band's tiles centre-out.
                const int band = ty / c->queue_bands;
                wedge = band % rto::kMaxQueues;
                bring = std::fabs((band + 0.5) * c->queue_bands - 0.5 - cy);
                bang = std::fabs(tx - cx) / (tx8 + 1.0);  // (< 1: orders inside a band; ring * 16 + ang stays monotone in the band)
                keyed.push_back({wedge, bring, bang, (uint32_t)(ty % c->queue_bands), code});
                continue;
            }
            keyed.push_back({wedge, bring, bang, spread((uint32_t)(tx % B)) | (spread((uint32_t)(ty % B)) << 1), code});
        }
    std::stable_sort(plain.begin(), plain.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    std::vector<uint32_t> order(plain.size());
    for (size_t i = 0; i < plain.size(); ++i) order[i] = plain[i].second;
    std::stable_sort(keyed.begin(), keyed.end(), [](const Keyed& a, const Keyed& b) {
        if (a.wedge != b.wedge) return a.wedge < b.wedge;
        const double ka = a.ring * 16.0 + a.ang, kb = b.ring * 16.0 + b.ang;
        if (ka != kb) return ka < kb;
        return a.morton < b.morton;
    });
    std::vector<uint32_t> worder(keyed.size());
    for (int k = 0; k <= rto::kMaxQueues; ++k) c->wedge_start[k] = 0;
    for (size_t i = 0; i < keyed.size(); ++i) {
        worder[i] = keyed[i].code;
        c->wedge_start[keyed[i].wedge + 1] = (int)i + 1;
    }
    for (int k = 1; k <= rto::kMaxQueues; ++k)  // empty wedges (tiny images) inherit the previous end
        if (c->wedge_start[k] < c->wedge_start[k - 1]) c->wedge_start[k] = c->wedge_start[k - 1];
    if (!c->tile_order && hipMalloc((void**)&c->tile_order, order.size() * 4) != hipSuccess)
        return set_err(RTO_E_HIP, "hipMalloc(tile_order) failed");
    if (!c->wedge_order && hipMalloc((void**)&c->wedge_order, worder.size() * 4) != hipSuccess)
        return set_err(RTO_E_HIP, "hipMalloc(tile_order) failed");
    if (hipMemcpy(c->tile_order, order.data(), order.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->wedge_order, worder.data(), worder.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        return set_err(RTO_E_HIP, "tile table upload failed");
    return RTO_OK;
}

int rto_ctx_create_batch(int width, int height, int frames, int device, rto_ctx** out) {
    if (width <= 0 || height <= 0 || !out) return set_err(RTO_E_INVALID, "rto_ctx_create: bad size");
    if (frames < 1 || frames > rto::kMaxBatch)
        return set_err(RTO_E_INVALID, "rto_ctx_create_batch: frames must be in 1.." + std::to_string(rto::kMaxBatch));
    if ((int64_t)width * height * 32 > 0x7fffffffLL)
        return set_err(RTO_E_INVALID, "rto_ctx_create: width*height*32 exceeds the int range of idx*SPP (volrend.cu:157)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_err(RTO_E_HIP, "no HIP device available (librto has no CPU fallback)");
    if (device < 0 || device >= ndev) return set_err(RTO_E_INVALID, "device index out of range");
    DeviceGuard guard(device);
    if (!guard.ok) return set_err(RTO_E_HIP, "hipSetDevice failed");
    auto c = new rto_ctx();
    c->device = device;
    c->width = width;
    c->height = height;
    c->frames = frames;
    c->lean_slot.assign((size_t)frames, 0);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            c->num_cus = prop.multiProcessorCount;
    }
    const size_t px = (size_t)width * height * frames;
    if (hipMalloc((void**)&c->aux, px * RTO_AUX_CHANNELS * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&c->noisy, px * 4 * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&c->image, px * 4 * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&c->rgba8, px * 4) != hipSuccess ||
        hipMalloc((void**)&c->queue, rto::kQueueWords * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void**)&c->d_frames, rto::kMaxBatch * sizeof(rto::FrameDesc)) != hipSuccess ||
        hipMemset(c->queue, 0, rto::kQueueWords * sizeof(unsigned long long)) != hipSuccess) {
        rto_ctx_free(c);
        return set_err(RTO_E_HIP, "hipMalloc(ctx buffers) failed");
    }
    (void)hipMemset(c->aux, 0, px * RTO_AUX_CHANNELS * sizeof(float));
    (void)hipMemset(c->noisy, 0, px * 4 * sizeof(float));
    (void)hipMemset(c->image, 0, px * 4 * sizeof(float));
    {
        int rc = build_tile_tables(c);
        if (rc != RTO_OK) {
            rto_ctx_free(c);
            return rc;
        }
    }
    pcg_seed(c->rng, 20230418ULL, 1);  // render_context.hpp:16
    for (int i = 0; i < 3; ++i) {
        if (hipEventCreate(&c->t_start[i]) != hipSuccess || hipEventCreate(&c->t_stop[i]) != hipSuccess) {
            rto_ctx_free(c);
            return set_err(RTO_E_HIP, "hipEventCreate failed");
        }
    }
    *out = c;
    return RTO_OK;
}

void rto_ctx_free(rto_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    if (c->aux) (void)hipFree(c->aux);
    if (c->noisy) (void)hipFree(c->noisy);
    if (c->image) (void)hipFree(c->image);
    if (c->rgba8) (void)hipFree(c->rgba8);
    if (c->jump) (void)hipFree(c->jump);
    if (c->queue) (void)hipFree(c->queue);
    for (void* p : {(void*)c->tile_mask, (void*)c->qlist, (void*)c->qscratch})
        if (p) (void)hipFree(p);
    if (c->d_frames) (void)hipFree(c->d_frames);
    if (c->hits) (void)hipFree(c->hits);
    if (c->tile_order) (void)hipFree(c->tile_order);
    if (c->wedge_order) (void)hipFree(c->wedge_order);
    for (hipEvent_t e : c->kt_ev) (void)hipEventDestroy(e);
    if (c->stats) (void)hipFree(c->stats);
    for (int i = 0; i < 3; ++i) {
        if (c->t_start[i]) (void)hipEventDestroy(c->t_start[i]);
        if (c->t_stop[i]) (void)hipEventDestroy(c->t_stop[i]);
    }
    delete c;
}

int rto_ctx_width(const rto_ctx* c) { return c ? c->width : 0; }
int rto_ctx_height(const rto_ctx* c) { return c ? c->height : 0; }
static size_t frame_px(const rto_ctx* c) { return (size_t)c->width * c->height; }
float* rto_ctx_aux(rto_ctx* c) { return c ? c->aux + (size_t)c->sel * RTO_AUX_CHANNELS * frame_px(c) : nullptr; }
float* rto_ctx_noisy(rto_ctx* c) { return c ? c->noisy + (size_t)c->sel * 4 * frame_px(c) : nullptr; }
float* rto_ctx_image(rto_ctx* c) { return c ? c->image + (size_t)c->sel * 4 * frame_px(c) : nullptr; }
int rto_ctx_frames(const rto_ctx* c) { return c ? c->frames : 0; }
int rto_ctx_selected_frame(const rto_ctx* c) { return c ? c->sel : 0; }
int rto_ctx_select_frame(rto_ctx* c, int frame) {
    if (!c || frame < 0 || frame >= c->frames) return set_err(RTO_E_INVALID, "rto_ctx_select_frame: frame out of range");
    c->sel = frame;
    return RTO_OK;
}

void rto_ctx_rng_seed(rto_ctx* c, uint64_t initstate, uint64_t initseq) {
    if (c) pcg_seed(c->rng, initstate, initseq);
}
void rto_ctx_rng_advance(rto_ctx* c, int64_t delta) {
    if (!c) return;
    const rto::PcgJumpEntry j = pcg_jump(c->rng.inc, (uint64_t)delta);
    c->rng.state = j.mult * c->rng.state + j.plus;
}
void rto_ctx_rng_set(rto_ctx* c, uint64_t state, uint64_t inc) {
    if (!c) return;
    c->rng.state = state;
    c->rng.inc = inc;
}
void rto_ctx_rng_get(const rto_ctx* c, uint64_t* state, uint64_t* inc) {
    if (!c) return;
    if (state) *state = c->rng.state;
    if (inc) *inc = c->rng.inc;
}

int rto_ctx_set_kernel(rto_ctx* c, int kernel) {
    if (!c || kernel < RTO_KERNEL_AUTO || kernel > RTO_KERNEL_FAST)
        return set_err(RTO_E_INVALID, "rto_ctx_set_kernel: bad argument");
    c->kernel = kernel;
    return RTO_OK;
}

#ifdef RTO_DBG_COUNTERS
extern "C" int rto_debug_read_queue(rto_ctx* c, uint64_t out[24]) {
    return hipMemcpy(out, c->queue, 24 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -4;
}
extern "C" int rto_debug_zero_queue(rto_ctx* c) { return hipMemset(c->queue + 2, 0, 48) == hipSuccess ? 0 : -4; }
extern "C" int rto_debug_shade_phases(uint64_t* out /* 2^19 waves x 8 words */, int reset) {
    return rto::debug_shade_phases((unsigned long long*)out, reset != 0) == hipSuccess ? 0 : -4;
}
#endif

// Host-only check of the two-level traversal image (no device needed): builds it for child[] (breadth-first node order) and
// walks it for n points given as 24-bit fixed-point coordinates, exactly as render_persist does (top grid of the wide image,
// two bits per axis per wide node, hit index -> leaf slot as flush_hits translates it).  out_level / out_slot / out_sigma:
// the leaf each point lies in.  tests/test_wide_image.py compares them with the plain walk over child[].
int rto_wide_image_probe(const int32_t* child, const uint16_t* sigma_bits, int64_t capacity, int max_depth, int top_levels,
                         const uint32_t* points, int64_t n, int32_t* out_level, int64_t* out_slot, uint16_t* out_sigma,
                         int64_t* out_wide_nodes) {
    if (!child || !sigma_bits || !points || !out_level || !out_slot || !out_sigma || capacity < 1 || n < 0)
        return set_err(RTO_E_INVALID, "rto_wide_image_probe: null argument");
    const int G = top_levels;
    WideImage wi;
    if (!build_wide_image(child, capacity, max_depth, G, [&](int64_t sl) { return sigma_bits[sl]; }, wi))
        return set_err(RTO_E_UNSUPPORTED, "rto_wide_image_probe: the tree has no two-level image (not breadth-first, too deep or too large)");
    if (out_wide_nodes) *out_wide_nodes = wi.n_wide;
    // the one-level image's words, for the hit-index translation (nodew: internal = child offset, leaf = tag)
    auto nodew_leaf = [&](int64_t slot) { return child[slot] == 0; };
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t ix = points[i * 3], iy = points[i * 3 + 1], iz = points[i * 3 + 2];
        // render_persist's walk: (node, off) = (0, 24 - G) at the grid, (node number, 22 - G - 2 p) at pair p below;
        // entry = ((node << b | x bits) << b | y bits) << b | z bits with b = node ? 2 : G bits per axis from bit `off` on
        uint32_t w = 0, u = 0, node = 0, off = 24u - (uint32_t)G;
        for (;;) {
            const uint32_t b = node ? 2u : (uint32_t)G, m = (1u << b) - 1u;
            u = (((node << b | ((ix >> off) & m)) << b | ((iy >> off) & m)) << b) | ((iz >> off) & m);
            w = wi.widew[u];
            if (rto::nodew_is_leaf(w)) break;
            node = w;  // internal: the node two levels down (from the grid: the level-G node's)
            off -= 2u;
        }
        // hit index -> leaf slot (render_kernels.hip wide_to_slot)
        int64_t slot;
        const uint32_t pad = wi.grid_nodes * 64u;
        if (u < pad) {
            slot = (int64_t)wi.gslot[u];
        } else {
            const uint32_t v = u - pad, wn = v >> 6, x2 = (v >> 4) & 3u, y2 = (v >> 2) & 3u, z2 = v & 3u;
            const uint32_t a = (x2 >> 1) << 2 | (y2 >> 1) << 1 | (z2 >> 1), b = (x2 & 1u) << 2 | (y2 & 1u) << 1 | (z2 & 1u);
            const int64_t N = wi.worig[wn];
            slot = nodew_leaf(N * 8 + a) ? N * 8 + a : (N + child[N * 8 + a]) * 8 + b;
        }
        out_level[i] = (int32_t)((w >> rto::kWideLevelShift) & 31u);
        out_slot[i] = slot;
        out_sigma[i] = (uint16_t)(w & 0xffffu);
    }
    return RTO_OK;
}

int rto_ctx_set_lean_outputs(rto_ctx* c, int level) {
    if (!c) return set_err(RTO_E_INVALID, "rto_ctx_set_lean_outputs: null context");
    if (level < 0 || level > 2) return set_err(RTO_E_INVALID, "rto_ctx_set_lean_outputs: level 0 (full), 1 (lean) or 2 (lean + sparse)");
    c->lean = level;
    return RTO_OK;
}

int rto_ctx_frames_lean_level(const rto_ctx* c, int first_slot, int n) {
    if (!c || n < 1 || first_slot < 0 || first_slot + n > c->frames) return 0;
    const int l0 = c->lean_slot[(size_t)first_slot];
    for (int i = first_slot + 1; i < first_slot + n; ++i)
        if (c->lean_slot[(size_t)i] != l0) return -1;
    return l0;
}

int rto_ctx_frames_are_lean(const rto_ctx* c, int first_slot, int n) {
    if (!c || n < 1 || first_slot < 0 || first_slot + n > c->frames) return 0;
    int lean = 0;
    for (int i = first_slot; i < first_slot + n; ++i) lean += c->lean_slot[(size_t)i] ? 1 : 0;
    return lean == n ? 1 : lean == 0 ? 0 : -1;  // -1: a mixed range -- no one route reads all of its slots correctly
}

int rto_ctx_set_tuning(rto_ctx* c, const char* key, int value) {
    if (!c || !key) return set_err(RTO_E_INVALID, "rto_ctx_set_tuning: null argument");
    const std::string k(key);
    if (k == "tile_order") {
        c->tile_order_on = value != 0;
    } else if (k == "xcd_queues") {
        c->xcd_queues = value != 0;
    } else if (k == "tile_major") {
        c->tile_major = value != 0;
    } else if (k == "tile_block") {
        if (value < 1 || value > 64) return set_err(RTO_E_INVALID, "tile_block must be 1..64");
        c->tile_block = value;
        DeviceGuard guard(c->device);
        if (hipDeviceSynchronize() != hipSuccess) return set_err(RTO_E_HIP, "hipDeviceSynchronize failed");
        return build_tile_tables(c);
    } else if (k == "queue_bands") {  // XCD queues by bands of `value` tile rows (0: angular wedges); same pixels
        if (value < 0 || value > 64) return set_err(RTO_E_INVALID, "queue_bands must be 0..64");
        c->queue_bands = value;
        DeviceGuard guard(c->device);
        if (hipDeviceSynchronize() != hipSuccess) return set_err(RTO_E_HIP, "hipDeviceSynchronize failed");
        return build_tile_tables(c);
    } else if (k == "refill") {
        c->refill = value;
    } else if (k == "wide_bits") {  // test hook (fast_path_for_spp): 0 = the real budget
        c->test_wide_bits = value;
    } else if (k == "cull") {  // empty-space culling of the batched path (1 = on; same pixels either way)
        c->cull_on = value != 0;
    } else if (k == "cull_single") {  // rto_launch_renderer's fast kernel skips the tiles no culling cell projects into (same pixels)
        c->cull_single = value != 0;
    } else if (k == "frame_via_batch") {  // rto_launch_renderer as a batch of one (culling + tile marks); same pixels
        c->frame_via_batch = value != 0;
    } else if (k == "blocks_per_cu") {  // occupancy of the persistent traversal kernel: 0 = what fits, else a cap (1..8)
        if (value < 0 || value > 8) return set_err(RTO_E_INVALID, "blocks_per_cu must be 0..8");
        c->occ.cap = value;
    } else if (k == "batch_fallback") {  // test hook: 1 = batched launches take the per-frame generic fallback (as a tree with too
        c->batch_fallback = value;       // many leaf slots for the SPP does), 2 = as if the device refused the traversal kernel's LDS
    } else if (k == "strip_rows") {
        if (value < 1) return set_err(RTO_E_INVALID, "strip_rows must be >= 1");
        c->strip_rows = value;
    } else {
        return set_err(RTO_E_INVALID, "unknown tuning key '" + k + "'");
    }
    return RTO_OK;
}

int rto_ctx_kernel_timing(rto_ctx* c, int enable) {
    if (!c) return set_err(RTO_E_INVALID, "rto_ctx_kernel_timing: null context");
    DeviceGuard guard(c->device);
    if (enable && c->kt_ev.empty()) {
        c->kt_ev.resize((size_t)kKtRing * 4);
        for (auto& e : c->kt_ev) HIP_TRY(hipEventCreate(&e));
    }
    c->kt_on = enable != 0;
    c->kt_count = 0;
    return RTO_OK;
}

int rto_ctx_kernel_timing_read(rto_ctx* c, float* traverse_ms, float* shade_ms, int* launches) {
    return rto_ctx_kernel_timing_read3(c, nullptr, traverse_ms, shade_ms, launches);
}

int rto_ctx_kernel_timing_read3(rto_ctx* c, float* raygen_ms, float* traverse_ms, float* shade_ms, int* launches) {
    if (!c) return set_err(RTO_E_INVALID, "rto_ctx_kernel_timing_read: null context");
    DeviceGuard guard(c->device);
    double g = 0, t = 0, s = 0;
    for (int i = 0; i < c->kt_count; ++i) {
        float r = 0, a = 0, b = 0;
        const hipEvent_t* e = &c->kt_ev[(size_t)i * 4];
        HIP_TRY(hipEventSynchronize(e[3]));
        HIP_TRY(hipEventElapsedTime(&r, e[0], e[1]));
        HIP_TRY(hipEventElapsedTime(&a, e[1], e[2]));
        HIP_TRY(hipEventElapsedTime(&b, e[2], e[3]));
        g += r;
        t += a;
        s += b;
    }
    if (raygen_ms) *raygen_ms = c->kt_count ? (float)(g / c->kt_count) : 0.f;
    if (traverse_ms) *traverse_ms = c->kt_count ? (float)(t / c->kt_count) : 0.f;
    if (shade_ms) *shade_ms = c->kt_count ? (float)(s / c->kt_count) : 0.f;
    if (launches) *launches = c->kt_count;
    c->kt_count = 0;
    return RTO_OK;
}

int rto_ctx_queue_stats(rto_ctx* c, int64_t* live_tile_slots, int64_t* all_tile_slots) {
    if (!c || !live_tile_slots || !all_tile_slots) return set_err(RTO_E_INVALID, "rto_ctx_queue_stats: null argument");
    if (!c->qscratch || c->last_n_queues < 1) return set_err(RTO_E_INVALID, "rto_ctx_queue_stats: no batched launch yet");
    DeviceGuard guard(c->device);
    HIP_TRY(hipDeviceSynchronize());
    uint32_t cnt[rto::kMaxQueues] = {0};
    HIP_TRY(hipMemcpy(cnt, c->qscratch + 2 * (size_t)c->q_chunks_cap, sizeof(cnt), hipMemcpyDeviceToHost));
    int64_t live = 0;
    for (int k = 0; k < c->last_n_queues; ++k) live += cnt[k];
    *live_tile_slots = live;
    *all_tile_slots = c->last_slots;
    return RTO_OK;
}

int rto_ctx_tile_marks(const rto_ctx* c, const uint32_t** marks, int* words_per_frame, int* first_slot, int* frames, float* background) {
    if (!c || !marks || !words_per_frame || !first_slot || !frames || !background)
        return set_err(RTO_E_INVALID, "rto_ctx_tile_marks: null argument");
    if (c->marks_n < 1 || !c->tile_mask) return set_err(RTO_E_INVALID, "rto_ctx_tile_marks: the last launch on this context was not a batched one");
    *marks = c->tile_mask;
    *words_per_frame = c->mask_words;
    *first_slot = c->marks_slot0;
    *frames = c->marks_n;
    *background = c->marks_bg;
    return RTO_OK;
}

int rto_ctx_enable_stats(rto_ctx* c, int enable) {
    if (!c) return set_err(RTO_E_INVALID, "rto_ctx_enable_stats: null context");
    DeviceGuard guard(c->device);
    if (enable && !c->stats) {
        HIP_TRY(hipMalloc((void**)&c->stats, rto::kStatsWords * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(c->stats, 0, rto::kStatsWords * sizeof(unsigned long long)));
    }
    c->stats_on = enable != 0;
    c->stats_marks = enable == 2;
    return RTO_OK;
}

int rto_ctx_get_march_stats(rto_ctx* c, void* stream_, uint64_t out[8], int reset) {
    if (!c || !out) return set_err(RTO_E_INVALID, "rto_ctx_get_march_stats: null argument");
    if (!c->stats) return set_err(RTO_E_INVALID, "rto_ctx_get_march_stats: counters were never enabled");
    DeviceGuard guard(c->device);
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipMemcpyAsync(out, c->stats + 6, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    if (reset) HIP_TRY(hipMemsetAsync(c->stats + 6, 0, 8 * sizeof(uint64_t), stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return RTO_OK;
}

int rto_ctx_get_stats(rto_ctx* c, void* stream_, uint64_t out[6], int reset) {
    if (!c || !out) return set_err(RTO_E_INVALID, "rto_ctx_get_stats: null argument");
    if (!c->stats) return set_err(RTO_E_INVALID, "rto_ctx_get_stats: counters were never enabled");
    DeviceGuard guard(c->device);
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipMemcpyAsync(out, c->stats, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    if (reset) HIP_TRY(hipMemsetAsync(c->stats, 0, 6 * sizeof(uint64_t), stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return RTO_OK;
}

static int launch_batch_at(const rto_tree* tree, const rto_camera* cams, const int64_t* rng_jumps, int n,
                           const rto_options* o, rto_ctx* ctx, void* stream_, int slot0);

int rto_launch_renderer(const rto_tree* tree, const rto_camera* cam, const rto_options* o, rto_ctx* ctx,
                        void* stream_) {
    if (!tree || !cam || !o || !ctx) return set_err(RTO_E_INVALID, "rto_launch_renderer: null argument");
    if (tree->quant) {  // codebook shading lives in the batched kernels: a batch of one into the selected slot
        if (ctx->kernel == RTO_KERNEL_GENERIC || ctx->stats_on)
            return set_err(RTO_E_UNSUPPORTED, "a quantised tree loaded with RTO_TREE_QUANT_DIRECT has no generic kernel / work counters");
        return launch_batch_at(tree, cam, nullptr, 1, o, ctx, stream_, ctx->sel);
    }
    if (!spp_supported(o->spp))  // volrend.cu:275-277
        return set_err(RTO_E_SPP, "spp == " + std::to_string(o->spp) + " not supported. (supported: 1,2,3,4,6,8,16,32)");
    if (cam->width != ctx->width || cam->height != ctx->height)
        return set_err(RTO_E_INVALID, "camera size does not match the render context");
    if (tree->device != ctx->device) return set_err(RTO_E_INVALID, "tree and context live on different devices");
    if (tree->d_recidx && !(o->sigma_thresh >= 0.f))
        return set_err(RTO_E_UNSUPPORTED, "a tree loaded with RTO_TREE_COMPACT_RECORDS keeps coefficient records for leaves of positive "
                                          "density only: sigma_thresh must be >= 0");
    if (o->enable_probe)
        return set_err(RTO_E_UNSUPPORTED, "enable_probe is a GUI feature (volrend.cu:100-134), not on the headless path");
    if (tree->dev.format == RTO_FMT_SG || tree->dev.format == RTO_FMT_ASG)
        return set_err(RTO_E_UNSUPPORTED, "SG/ASG bases are untested upstream (lumisphere.hpp:14-37) and not built");
    if (!(cam->fx != 0.f) || !(cam->fy != 0.f)) return set_err(RTO_E_INVALID, "camera focal length is zero");

    int kernel = ctx->kernel;
    rto::TreeDev tdev;
    const bool fast_here = fast_path_for_spp(tree, o->spp, ctx->test_wide_bits, &tdev) != 0;
    if (kernel == RTO_KERNEL_AUTO) kernel = fast_here ? RTO_KERNEL_FAST : RTO_KERNEL_GENERIC;
    if (kernel == RTO_KERNEL_FAST && !fast_here)
        return set_err(RTO_E_UNSUPPORTED, "fast kernel needs an N == 2 tree of depth <= 24 whose leaf slots fit 31 - ceil(log2 spp) bits "
                                          "(2^28 slots at spp <= 8, 2^26 at spp 32: a hit entry is {valid bit, count - 1, slot})");

    if (kernel == RTO_KERNEL_FAST && ctx->frame_via_batch && ctx->kernel == RTO_KERNEL_AUTO && !ctx->stats_on)
        return launch_batch_at(tree, cam, nullptr, 1, o, ctx, stream_, ctx->sel);

    DeviceGuard guard(ctx->device);
    if (!guard.ok) return set_err(RTO_E_HIP, "hipSetDevice failed");
    hipStream_t stream = (hipStream_t)stream_;
    if (kernel == RTO_KERNEL_FAST) {
        int rc = ensure_jump_table(ctx, stream);
        if (rc != RTO_OK) return rc;
    } else {
        int rc = ensure_reference_arrays(tree);  // (a no-op unless the upload released them)
        if (rc != RTO_OK) return rc;
    }

    rto::CamDev cd;
    cd.width = cam->width;
    cd.height = cam->height;
    cd.fx = cam->fx;
    cd.fy = cam->fy;
    std::memcpy(cd.transform, cam->transform, sizeof(cd.transform));
    const rto::OptDev od = make_opt_dev(o);
    rto::FrameOut fo = {};
    fo.aux = rto_ctx_aux(ctx);
    fo.image = o->denoise ? rto_ctx_noisy(ctx) : rto_ctx_image(ctx);  // volrend.cu:206
    fo.stats = nullptr;
    fo.stat_marks = nullptr;
    fo.stat_mask_words = 0;
    fo.cull_marks = nullptr;
    fo.cull_mask_words = 0;
    bool keep_marks = false;
    // Empty-space culling for the single-frame kernel as well (round 4, VERDICT r3 task 6): one small kernel projects the
    // tree's culling cells into this camera (same bound and premises as the batched path, see launch_batch_at), and the
    // waves of unmarked 8x8 tiles write the background without ray set-up, threshold draws or marching.  The marks stay
    // on the context for the denoise stage (rto_ctx_tile_marks), like after a batched launch.
    const bool cull_one = kernel == RTO_KERNEL_FAST && !ctx->stats_on && ctx->cull_on && ctx->cull_single && tree->dev.occ_cells &&
                          o->sigma_thresh >= 0.f && !(tree->dev.ndc_width > 0.f);
    if (cull_one) {
        const int tiles = ((ctx->width + 7) / 8) * ((ctx->height + 7) / 8);
        const int mask_words = (tiles + 31) / 32 + 1;
        if (!ctx->tile_mask || ctx->mask_words != mask_words) {
            if (ctx->tile_mask) {
                HIP_TRY(hipDeviceSynchronize());
                for (void* p : {(void*)ctx->tile_mask, (void*)ctx->qlist, (void*)ctx->qscratch}) HIP_TRY(hipFree(p));
                ctx->tile_mask = ctx->qlist = ctx->qscratch = nullptr;
            }
            ctx->q_chunks_cap = 0;  // (a batched launch sizes its queue lists itself)
            HIP_TRY(hipMalloc((void**)&ctx->tile_mask, (size_t)ctx->frames * mask_words * sizeof(uint32_t)));
            ctx->mask_words = mask_words;
        }
        HIP_TRY(rto::launch_mark_tiles_one(tdev, cd, ctx->tile_mask, mask_words, stream));
        fo.cull_marks = ctx->tile_mask;
        fo.cull_mask_words = mask_words;
    }
    if (ctx->stats_on) {
        if (kernel != RTO_KERNEL_FAST) return set_err(RTO_E_UNSUPPORTED, "work counters need the fast kernel");
        fo.stats = ctx->stats;
        if (ctx->stats_marks) {  // the selected slot's marks of the last batched launch (the caller re-renders that frame)
            if (ctx->marks_n < 1 || !ctx->tile_mask || ctx->sel < ctx->marks_slot0 || ctx->sel >= ctx->marks_slot0 + ctx->marks_n)
                return set_err(RTO_E_INVALID, "rto_ctx_enable_stats(2): the selected frame slot holds no tile marks of a batched launch");
            fo.stat_marks = ctx->tile_mask + (size_t)(ctx->sel - ctx->marks_slot0) * ctx->mask_words;
            fo.stat_mask_words = ctx->mask_words;
            keep_marks = true;  // (the counting launch re-renders the frame the marks describe)
        }
    }

    if (!keep_marks) ctx->marks_n = 0;  // (the generic kernel and the counting instantiation mark no tiles)
    ctx->lean_slot[(size_t)ctx->sel] = 0;  // (a single frame has full outputs; the other slots keep what they hold)
    // (the generic kernel reads tree->dev itself: child[] / data[] may just have been rebuilt by ensure_reference_arrays)
    hipError_t e = rto::launch_render(kernel, o->spp, kernel == RTO_KERNEL_FAST ? tdev : tree->dev, cd, od, ctx->rng, ctx->jump, fo,
                                      ctx->strip_rows, stream);
    if (e != hipSuccess) return set_err(RTO_E_HIP, std::string("render launch failed: ") + hipGetErrorString(e));
    if (cull_one) {  // the selected slot's marks, for rto_ctx_tile_marks / the culled denoise stage
        ctx->marks_n = 1;
        ctx->marks_slot0 = ctx->sel;
        ctx->marks_bg = o->background_brightness;
    }
    return RTO_OK;
}

int rto_launch_renderer_batch(const rto_tree* tree, const rto_camera* cams, const int64_t* rng_jumps, int n,
                              const rto_options* o, rto_ctx* ctx, void* stream_) {
    return launch_batch_at(tree, cams, rng_jumps, n, o, ctx, stream_, 0);
}

// The frames of a batched call rendered one by one with the generic kernel (render_generic: any N, any depth, any slot
// count): what rto_launch_renderer_batch does for a tree the batched kernels cannot take.  The generic kernel reads the
// reference arrays child[] / data[], which a dense SH9 / SH16 upload released: they are rebuilt first.
// No tile marks exist afterwards (marks_n stays 0: the denoise stage must not fill "culled" tiles from an older launch).
static int generic_frames(const rto_tree* tree, const rto_camera* cams, const int64_t* rng_jumps, int n, const rto_options* o,
                          rto_ctx* ctx, void* stream_, int slot0) {
    ctx->marks_n = 0;
    if (tree->quant)
        return set_err(RTO_E_UNSUPPORTED, "a quantised tree kept quantised cannot take the generic kernel (too many leaf slots for the "
                                          "batched kernels at this spp, or the traversal kernel's LDS was refused)");
    int rc = ensure_reference_arrays(tree);
    if (rc != RTO_OK) return rc;
    const rto::OptDev od = make_opt_dev(o);
    const size_t px = frame_px(ctx);
    for (int f = 0; f < n; ++f) {
        if (cams[f].width != ctx->width || cams[f].height != ctx->height)
            return set_err(RTO_E_INVALID, "camera size does not match the render context");
        if (!(cams[f].fx != 0.f) || !(cams[f].fy != 0.f)) return set_err(RTO_E_INVALID, "camera focal length is zero");
    }
    for (int f = 0; f < n; ++f) {
        rto::CamDev cd;
        cd.width = cams[f].width;
        cd.height = cams[f].height;
        cd.fx = cams[f].fx;
        cd.fy = cams[f].fy;
        std::memcpy(cd.transform, cams[f].transform, sizeof(cd.transform));
        const int64_t jumps = rng_jumps ? rng_jumps[f] : (int64_t)f;
        const rto::PcgJumpEntry j = pcg_jump(ctx->rng.inc, (uint64_t)jumps << 32);
        rto::Pcg32 rng = ctx->rng;
        rng.state = j.mult * ctx->rng.state + j.plus;
        const size_t slot = (size_t)(slot0 + f);
        rto::FrameOut fo = {};
        fo.aux = ctx->aux + slot * RTO_AUX_CHANNELS * px;
        fo.image = (o->denoise ? ctx->noisy : ctx->image) + slot * 4 * px;
        fo.stats = nullptr;
        fo.stat_marks = nullptr;
        fo.stat_mask_words = 0;
        hipError_t e = rto::launch_render(RTO_KERNEL_GENERIC, o->spp, tree->dev, cd, od, rng, ctx->jump, fo, ctx->strip_rows,
                                          (hipStream_t)stream_);
        if (e != hipSuccess) return set_err(RTO_E_HIP, std::string("render launch failed: ") + hipGetErrorString(e));
    }
    return RTO_OK;
}

// frames 0..n-1 of the batch land in context slots slot0..slot0+n-1
static int launch_batch_at(const rto_tree* tree, const rto_camera* cams, const int64_t* rng_jumps, int n,
                           const rto_options* o, rto_ctx* ctx, void* stream_, int slot0) {
    if (!tree || !cams || !o || !ctx) return set_err(RTO_E_INVALID, "rto_launch_renderer_batch: null argument");
    if (n < 1 || slot0 < 0 || slot0 + n > ctx->frames)
        return set_err(RTO_E_INVALID, "rto_launch_renderer_batch: n exceeds the context's frame slots");
    if (!spp_supported(o->spp))
        return set_err(RTO_E_SPP, "spp == " + std::to_string(o->spp) + " not supported. (supported: 1,2,3,4,6,8,16,32)");
    if (tree->device != ctx->device) return set_err(RTO_E_INVALID, "tree and context live on different devices");
    if (tree->d_recidx && !(o->sigma_thresh >= 0.f))
        return set_err(RTO_E_UNSUPPORTED, "a tree loaded with RTO_TREE_COMPACT_RECORDS keeps coefficient records for leaves of positive "
                                          "density only: sigma_thresh must be >= 0");
    if (o->enable_probe) return set_err(RTO_E_UNSUPPORTED, "enable_probe is a GUI feature, not on the headless path");
    if (tree->dev.format == RTO_FMT_SG || tree->dev.format == RTO_FMT_ASG)
        return set_err(RTO_E_UNSUPPORTED, "SG/ASG bases are untested upstream and not built");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return set_err(RTO_E_HIP, "hipSetDevice failed");
    ctx->marks_n = 0;  // whatever happens below, the tile marks of an earlier launch no longer describe this context's frames
    for (int f = 0; f < n; ++f) ctx->lean_slot[(size_t)(slot0 + f)] = 0;  // (set again below if this launch stores lean; the generic fallback stores full outputs)
    rto::TreeDev tdev;
    if (fast_path_for_spp(tree, o->spp, ctx->test_wide_bits, &tdev) == 0 || ctx->batch_fallback == 1) {
        // No traversal image (N != 2, depth > 24, >= 2^29 leaf slots: the top-grid entry's budget) or more slots than a
        // hit-list entry can name at this SPP (2^28 at spp <= 8, 2^26 at spp 32): the same frames, one launch of the
        // generic kernel each -- same images, without the batching gain.
        return generic_frames(tree, cams, rng_jumps, n, o, ctx, stream_, slot0);
    }
    // the traversal kernel addresses the hand-off buffer with 32-bit offsets (frame * spp * pixels + pixel)
    if ((uint64_t)(slot0 + n) * (uint64_t)o->spp * (uint64_t)frame_px(ctx) > 0xffffffffULL)
        return set_err(RTO_E_UNSUPPORTED, "frames x spp x pixels exceeds 2^32 hit-list entries: render fewer frames per launch");
    if (ctx->hits_spp < o->spp) {  // grow the hit-list buffer (first use, or a larger spp); stream-ordered free
        if (ctx->hits) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(ctx->hits));
            ctx->hits = nullptr;
        }
        HIP_TRY(hipMalloc((void**)&ctx->hits, (size_t)ctx->frames * o->spp * frame_px(ctx) * sizeof(uint32_t)));
        ctx->hits_spp = o->spp;
    }
    rto::FrameBatch fb;
    std::memset(&fb, 0, sizeof(fb));
    fb.n = n;
    fb.width = ctx->width;
    fb.height = ctx->height;
    {
        const int tiles = ((ctx->width + 7) / 8) * ((ctx->height + 7) / 8);
        fb.tile_major = ctx->tile_major ? 1 : 0;
        if (ctx->tile_order_on && ctx->xcd_queues) {  // one queue per XCD over an image wedge each
            fb.tile_order = ctx->wedge_order;
            fb.n_queues = rto::kMaxQueues;
            for (int k = 0; k <= rto::kMaxQueues; ++k) fb.qstart[k] = ctx->wedge_start[k];
        } else {  // one queue over whole frames (centre-out or row-major tiles)
            fb.tile_order = ctx->tile_order_on ? ctx->tile_order : nullptr;
            fb.n_queues = 1;
            fb.qstart[0] = 0;
            for (int k = 1; k <= rto::kMaxQueues; ++k) fb.qstart[k] = tiles;
        }
    }
    {   // tile marks + queue lists
        const int tiles_x = (ctx->width + 7) / 8, tiles_y = (ctx->height + 7) / 8, tiles = tiles_x * tiles_y;
        if (tiles_x > 1023 || tiles_y > 1023)
            return set_err(RTO_E_UNSUPPORTED, "the batched path packs tile coordinates into 10 bits: frames up to 8184 x 8184 pixels");
        const int mask_words = (tiles + 31) / 32 + 1;
        fb.qchunk[0] = 0;
        for (int k = 0; k < rto::kMaxQueues; ++k) {
            const int64_t slots = k < fb.n_queues ? (int64_t)(fb.qstart[k + 1] - fb.qstart[k]) * n : 0;
            fb.qchunk[k + 1] = fb.qchunk[k] + (int)((slots + rto::kQueueChunk - 1) / rto::kQueueChunk);
        }
        const int chunks_cap = (int)(((int64_t)tiles * ctx->frames + rto::kQueueChunk - 1) / rto::kQueueChunk) + rto::kMaxQueues;
        if (!ctx->tile_mask || ctx->mask_words != mask_words || ctx->q_chunks_cap < chunks_cap) {
            if (ctx->tile_mask) {
                HIP_TRY(hipDeviceSynchronize());
                for (void* p : {(void*)ctx->tile_mask, (void*)ctx->qlist, (void*)ctx->qscratch}) HIP_TRY(hipFree(p));
                ctx->tile_mask = ctx->qlist = ctx->qscratch = nullptr;
            }
            HIP_TRY(hipMalloc((void**)&ctx->tile_mask, (size_t)ctx->frames * mask_words * sizeof(uint32_t)));
            HIP_TRY(hipMalloc((void**)&ctx->qlist, (size_t)tiles * ctx->frames * sizeof(uint32_t)));
            HIP_TRY(hipMalloc((void**)&ctx->qscratch, ((size_t)2 * chunks_cap + rto::kMaxQueues) * sizeof(uint32_t)));
            ctx->mask_words = mask_words;
            ctx->q_chunks_cap = chunks_cap;
        }
        fb.tile_mask = ctx->tile_mask;
        fb.mask_words = mask_words;
        fb.qlist = ctx->qlist;
        fb.chunk_count = ctx->qscratch;
        fb.chunk_base = ctx->qscratch + ctx->q_chunks_cap;
        fb.qcount = ctx->qscratch + 2 * (size_t)ctx->q_chunks_cap;
    }
    // Rays that provably never meet density are not marched (see mark_tiles_kernel): needs the tree's culling cells, a
    // non-negative density threshold (a hit needs sigma > sigma_thresh, rt_core.cuh:252) and straight world-space rays
    // (the NDC warp of LLFF scenes bends them: maybe_world2ndc, volrend.cu:35-56)
    ctx->last_n_queues = fb.n_queues;
    ctx->last_slots = (int64_t)((ctx->width + 7) / 8) * ((ctx->height + 7) / 8) * n;
    const bool cull = ctx->cull_on && tree->dev.occ_cells && o->sigma_thresh >= 0.f && !(tree->dev.ndc_width > 0.f);
    const size_t px = frame_px(ctx);
    rto::FrameDesc frames[rto::kMaxBatch];
    fb.f = ctx->d_frames;
    fb.lean = o->denoise ? ctx->lean : 0;
    for (int f = 0; f < n; ++f) {
        if (cams[f].width != ctx->width || cams[f].height != ctx->height)
            return set_err(RTO_E_INVALID, "camera size does not match the render context");
        if (!(cams[f].fx != 0.f) || !(cams[f].fy != 0.f)) return set_err(RTO_E_INVALID, "camera focal length is zero");
        rto::FrameDesc& d = frames[f];
        d.fx = cams[f].fx;
        d.fy = cams[f].fy;
        std::memcpy(d.transform, cams[f].transform, sizeof(d.transform));
        // frame f = what the f-th of n sequential launch_renderer calls would see with
        // ctx.rng.advance() in between (main_headless.cpp:494-506), or an explicit jump count
        const int64_t jumps = rng_jumps ? rng_jumps[f] : (int64_t)f;
        const rto::PcgJumpEntry j = pcg_jump(ctx->rng.inc, (uint64_t)jumps << 32);
        d.rng_state = j.mult * ctx->rng.state + j.plus;
        d.rng_inc = ctx->rng.inc;
        const size_t slot = (size_t)(slot0 + f);
        d.aux = ctx->aux + slot * RTO_AUX_CHANNELS * px;
        d.image = (o->denoise ? ctx->noisy : ctx->image) + slot * 4 * px;
        d.hits = ctx->hits + slot * o->spp * px;
    }
    hipStream_t stream = (hipStream_t)stream_;
    int rc = ensure_jump_table(ctx, stream);
    if (rc != RTO_OK) return rc;
    const rto::OptDev od = make_opt_dev(o);
    HIP_TRY(rto::launch_write_frames(frames, n, ctx->d_frames, stream));
    hipEvent_t* ev = nullptr;
    if (ctx->kt_on && ctx->kt_count < kKtRing) ev = &ctx->kt_ev[(size_t)ctx->kt_count++ * 4];
    ctx->occ.force_lds_refusal = ctx->batch_fallback == 2;
    hipError_t e = rto::launch_render_batch(o->spp, tdev, od, fb, ctx->jump, ctx->queue,
                                            ctx->hits + (size_t)slot0 * o->spp * px,  // = fb.f[0].hits: the kernel indexes frames from here
                                            ctx->num_cus, ctx->refill, cull, &ctx->occ, ev, stream);
    if (ctx->occ.lds_refused) {
        if (ev) --ctx->kt_count;  //  nothing was launched, nothing recorded the slot's events: give it back
        // the device does not grant a workgroup the LDS the traversal kernel needs for this depth x SPP x frame count
        // ((max_depth + 1 - top_levels + spp + 1) KB for the ancestor stack and thresholds + 56 B per frame); nothing but
        // the frame table was written so far: the same frames through the generic kernel, frame by frame
        return generic_frames(tree, cams, rng_jumps, n, o, ctx, stream_, slot0);
    }
    if (e != hipSuccess) return set_err(RTO_E_HIP, std::string("batched render launch failed: ") + hipGetErrorString(e));
    ctx->marks_n = n;
    ctx->marks_slot0 = slot0;
    ctx->marks_bg = o->background_brightness;
    if (fb.lean)
        for (int f = 0; f < n; ++f) ctx->lean_slot[(size_t)(slot0 + f)] = (uint8_t)fb.lean;
    return RTO_OK;
}

int rto_filtering_batch(void* stream, const float* weight_map, const float* guidance_map, int L, int H, int W, int n,
                        const float* img_in, float* img_out) {
    if (!weight_map || !guidance_map || !img_in || !img_out || H <= 0 || W <= 0 || n < 1)
        return set_err(RTO_E_INVALID, "rto_filtering: null pointer or bad size");
    if (L < 1 || L > 6)  // filtering.cu:362-366
        return set_err(RTO_E_INVALID, "Kernel size == " + std::to_string(L * 2 + 1) + " not supported.");
    if (img_in == img_out) return set_err(RTO_E_INVALID, "rto_filtering: img_in and img_out must differ");
    const int pdev_ = device_of(img_out);
    if (pdev_ < 0) return set_err(RTO_E_INVALID, "filtering: the output pointer is not device memory");
    DeviceGuard guard(pdev_);
    if (!guard.ok) return set_err(RTO_E_HIP, "hipSetDevice failed");
    hipError_t e = rto::launch_filter(weight_map, guidance_map, L, H, W, n, img_in, img_out, (hipStream_t)stream);
    if (e != hipSuccess) return set_err(RTO_E_HIP, std::string("filter launch failed: ") + hipGetErrorString(e));
    return RTO_OK;
}

int rto_filtering_batch_mode(void* stream, const float* weight_map, const float* guidance_map, int L, int H, int W, int n,
                             const float* img_in, float* img_out, int mode) {
    if (mode == RTO_FILTER_EXACT) return rto_filtering_batch(stream, weight_map, guidance_map, L, H, W, n, img_in, img_out);
    if (mode != RTO_FILTER_FACTORISED) return set_err(RTO_E_INVALID, "rto_filtering_batch_mode: unknown mode");
    if (!weight_map || !guidance_map || !img_in || !img_out || H <= 0 || W <= 0 || n < 1)
        return set_err(RTO_E_INVALID, "rto_filtering: null pointer or bad size");
    if (L < 1 || L > 6) return set_err(RTO_E_INVALID, "Kernel size == " + std::to_string(L * 2 + 1) + " not supported.");
    if (img_in == img_out) return set_err(RTO_E_INVALID, "rto_filtering: img_in and img_out must differ");
    const int pdev_ = device_of(img_out);
    if (pdev_ < 0) return set_err(RTO_E_INVALID, "filtering: the output pointer is not device memory");
    DeviceGuard guard(pdev_);
    if (!guard.ok) return set_err(RTO_E_HIP, "hipSetDevice failed");
    hipError_t e = rto::launch_filter_fast(weight_map, guidance_map, L, H, W, n, img_in, img_out, (hipStream_t)stream);
    if (e != hipSuccess) return set_err(RTO_E_HIP, std::string("filter launch failed: ") + hipGetErrorString(e));
    return RTO_OK;
}

int rto_filtering_train_forward(void* stream, const float* weight_map, const float* guidance_map, int L, int H, int W,
                                int n, const float* img_in, float* img_out, float* rgb_filtered, float* max_map,
                                float* inv_kernel_sum) {
    if (!weight_map || !guidance_map || !img_in || !img_out || !rgb_filtered || !max_map || !inv_kernel_sum || H <= 0 ||
        W <= 0 || n < 1)
        return set_err(RTO_E_INVALID, "rto_filtering_train_forward: null pointer or bad size");
    if (L < 1 || L > 6) return set_err(RTO_E_INVALID, "Kernel size == " + std::to_string(L * 2 + 1) + " not supported.");
    if (img_in == img_out) return set_err(RTO_E_INVALID, "rto_filtering_train_forward: img_in and img_out must differ");
    const int pdev_ = device_of(img_out);
    if (pdev_ < 0) return set_err(RTO_E_INVALID, "filtering: the output pointer is not device memory");
    DeviceGuard guard(pdev_);
    if (!guard.ok) return set_err(RTO_E_HIP, "hipSetDevice failed");
    hipError_t e = rto::launch_filter_train(weight_map, guidance_map, L, H, W, n, img_in, img_out, rgb_filtered, max_map,
                                            inv_kernel_sum, (hipStream_t)stream);
    if (e != hipSuccess) return set_err(RTO_E_HIP, std::string("filter launch failed: ") + hipGetErrorString(e));
    return RTO_OK;
}

int rto_filtering_backward(void* stream, const float* grad_output, const float* img_in, const float* weight_map,
                           const float* guidance_map, const float* rgb_filtered, const float* max_map,
                           const float* inv_kernel_sum, int L, int H, int W, int n, float* grad_weight,
                           float* grad_guidance) {
    if (!grad_output || !img_in || !weight_map || !guidance_map || !rgb_filtered || !max_map || !inv_kernel_sum ||
        !grad_weight || !grad_guidance || H <= 0 || W <= 0 || n < 1)
        return set_err(RTO_E_INVALID, "rto_filtering_backward: null pointer or bad size");
    if (L < 1 || L > 6) return set_err(RTO_E_INVALID, "Kernel size == " + std::to_string(L * 2 + 1) + " not supported.");
    const int pdev_ = device_of(grad_guidance);
    if (pdev_ < 0) return set_err(RTO_E_INVALID, "filtering: the output pointer is not device memory");
    DeviceGuard guard(pdev_);
    if (!guard.ok) return set_err(RTO_E_HIP, "hipSetDevice failed");
    hipError_t e = rto::launch_filter_backward(grad_output, img_in, weight_map, guidance_map, rgb_filtered, max_map,
                                               inv_kernel_sum, L, H, W, n, grad_weight, grad_guidance, (hipStream_t)stream);
    if (e != hipSuccess) return set_err(RTO_E_HIP, std::string("filter backward launch failed: ") + hipGetErrorString(e));
    return RTO_OK;
}

int rto_filtering(void* stream, const float* weight_map, const float* guidance_map, int L, int H, int W,
                  const float* img_in, float* img_out) {
    return rto_filtering_batch(stream, weight_map, guidance_map, L, H, W, 1, img_in, img_out);
}

int rto_ctx_filtering(rto_ctx* c, void* stream, const float* weight_map, const float* guidance_map, int L) {
    if (!c) return set_err(RTO_E_INVALID, "rto_ctx_filtering: null context");
    DeviceGuard guard(c->device);
    return rto_filtering(stream, weight_map, guidance_map, L, c->height, c->width, rto_ctx_noisy(c), rto_ctx_image(c));
}

// The noisy image of a SPARSE lean frame holds nothing in the tiles its launch left unmarked: they are the background (colour =
// the launch's background brightness, alpha 0 -- a lean frame's alpha is aux plane 3).  Filled in on the host after the copy,
// from the marks the launch left on the device; T = float (x4 per pixel) or uint8_t (the truncated bytes, main_headless.cpp:535-538).
}  // extern "C"
template <class T>
static int fill_unmarked_tiles(rto_ctx* c, hipStream_t stream, T* host, T colour) {
    if (c->lean_slot[(size_t)c->sel] != 2) return RTO_OK;
    if (c->marks_n < 1 || !c->tile_mask || c->sel < c->marks_slot0 || c->sel >= c->marks_slot0 + c->marks_n)
        return set_err(RTO_E_INVALID, "the selected slot holds a sparse lean frame whose tile marks a later launch replaced: its noisy image "
                                      "cannot be completed");
    std::vector<uint32_t> m((size_t)c->mask_words);
    HIP_TRY(hipMemcpyAsync(m.data(), c->tile_mask + (size_t)(c->sel - c->marks_slot0) * c->mask_words, m.size() * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (m.back() & 1u) return RTO_OK;  // keep-all frame: every tile was rendered
    const int tx_n = (c->width + 7) / 8, ty_n = (c->height + 7) / 8;
    for (int ty = 0; ty < ty_n; ++ty)
        for (int tx = 0; tx < tx_n; ++tx) {
            const uint32_t t = (uint32_t)(ty * tx_n + tx);
            if ((m[t >> 5] >> (t & 31u)) & 1u) continue;
            for (int y = ty * 8; y < std::min(ty * 8 + 8, c->height); ++y)
                for (int x = tx * 8; x < std::min(tx * 8 + 8, c->width); ++x) {
                    T* p = host + ((size_t)y * c->width + x) * 4;
                    p[0] = p[1] = p[2] = colour;
                    p[3] = T(0);
                }
        }
    return RTO_OK;
}
extern "C" {

int rto_ctx_download_rgba8(rto_ctx* c, void* stream_, int which, uint8_t* host_out) {
    if (!c || !host_out) return set_err(RTO_E_INVALID, "rto_ctx_download_rgba8: null argument");
    DeviceGuard guard(c->device);
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t px = (int64_t)c->width * c->height;
    HIP_TRY(rto::launch_rgba8(which ? rto_ctx_noisy(c) : rto_ctx_image(c), c->rgba8, px, stream));
    HIP_TRY(hipMemcpyAsync(host_out, c->rgba8, (size_t)px * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (which) return fill_unmarked_tiles<uint8_t>(c, stream, host_out, (uint8_t)(c->marks_bg * 255));
    return RTO_OK;
}

int rto_ctx_download_image(rto_ctx* c, void* stream_, int which, float* host_out) {
    if (!c || !host_out) return set_err(RTO_E_INVALID, "rto_ctx_download_image: null argument");
    DeviceGuard guard(c->device);
    hipStream_t stream = (hipStream_t)stream_;
    const size_t bytes = (size_t)c->width * c->height * 4 * sizeof(float);
    HIP_TRY(hipMemcpyAsync(host_out, which ? rto_ctx_noisy(c) : rto_ctx_image(c), bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (which) return fill_unmarked_tiles<float>(c, stream, host_out, c->marks_bg);
    return RTO_OK;
}

int rto_ctx_download_aux(rto_ctx* c, void* stream_, float* host_out) {
    if (!c || !host_out) return set_err(RTO_E_INVALID, "rto_ctx_download_aux: null argument");
    DeviceGuard guard(c->device);
    hipStream_t stream = (hipStream_t)stream_;
    const size_t bytes = (size_t)c->width * c->height * RTO_AUX_CHANNELS * sizeof(float);
    HIP_TRY(hipMemcpyAsync(host_out, rto_ctx_aux(c), bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return RTO_OK;
}

// ---- Timer (render_context.hpp:122-213) ----
int rto_timer_reset(rto_ctx* c, void* stream) {
    if (!c) return set_err(RTO_E_INVALID, "rto_timer_reset: null context");
    c->t_stream = (hipStream_t)stream;
    c->t_cnt = 0;
    for (int i = 0; i < 3; ++i) {
        c->t_sum[i] = 0;
        c->t_used[i] = false;
    }
    return RTO_OK;
}
int rto_timer_start(rto_ctx* c, int which) {
    if (!c || which < 0 || which > 2) return set_err(RTO_E_INVALID, "rto_timer_start: bad argument");
    DeviceGuard guard(c->device);
    HIP_TRY(hipEventRecord(c->t_start[which], c->t_stream));
    return RTO_OK;
}
int rto_timer_stop(rto_ctx* c, int which) {
    if (!c || which < 0 || which > 2) return set_err(RTO_E_INVALID, "rto_timer_stop: bad argument");
    DeviceGuard guard(c->device);
    HIP_TRY(hipEventRecord(c->t_stop[which], c->t_stream));
    c->t_used[which] = true;
    return RTO_OK;
}
int rto_timer_record(rto_ctx* c, int denoise) {
    if (!c) return set_err(RTO_E_INVALID, "rto_timer_record: null context");
    DeviceGuard guard(c->device);
    const int last = denoise ? RTO_T_FILTER : RTO_T_RENDER;
    if (!c->t_used[last]) return set_err(RTO_E_INVALID, "rto_timer_record: the closing event was never recorded");
    HIP_TRY(hipEventSynchronize(c->t_stop[last]));
    c->t_cnt++;
    for (int i = 0; i < 3; ++i) {
        // the reference reads all three pairs every frame; buckets that never ran stay at 0 here
        if (!c->t_used[i] || (!denoise && i != RTO_T_RENDER)) continue;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->t_start[i], c->t_stop[i]));
        c->t_sum[i] += ms;
    }
    return RTO_OK;
}
int rto_timer_report(const rto_ctx* c, float ms_out[3], float* fps_out, int* frames_out) {
    if (!c) return set_err(RTO_E_INVALID, "rto_timer_report: null context");
    float all = 0;
    for (int i = 0; i < 3; ++i) {
        const float t = c->t_cnt ? c->t_sum[i] / c->t_cnt : 0.f;
        if (ms_out) ms_out[i] = t;
        all += t;
    }
    if (fps_out) *fps_out = all > 0 ? 1000.f / all : 0.f;
    if (frames_out) *frames_out = c->t_cnt;
    return RTO_OK;
}

}  // extern "C"
