// rto_kernel_types.h -- POD kernel-argument views (the role of the reference's
// internal/data_spec.hpp:11-52 TreeSpec / CameraSpec), passed to the kernels BY VALUE.
#pragma once
#include <stdint.h>

#include "rto_device_math.h"

namespace rto {

#define RTO_BASIS_MAX_DEV 25  // render_options.hpp:7 VOLREND_GLOBAL_BASIS_MAX

// Leaf tag of the traversal image `nodew` (see build_nodew_kernel in render_kernels.hip):
//   internal slot: the reference's child[] value (relative node offset, |v| < 2^30)
//   leaf slot:     0x80000000 | fp16 bits of the slot's sigma  -> top two bits are 0b10
constexpr uint32_t kLeafTag = 0x80000000u;
// A leaf word of the TWO-LEVEL image also carries its leaf's level (< 32), at the bits a float's exponent field starts at:
// the march step's 2^(level + c) factors are then one integer add / subtract on (word & kWideLevelMask) -- no field extract
constexpr int kWideLevelShift = 23;
constexpr uint32_t kWideLevelMask = 31u << kWideLevelShift;
constexpr int kQueueChunk = 256;  // tile slots per workgroup of the queue compaction
constexpr int kOccLevel = 7;  // finest cube of the culling cells: 2^-7 of the volume (6 pixels across at 800 x 800)
constexpr uint32_t kNoRecord = 0xffffffffu;  // TreeDev::recidx entry of a slot without a coefficient record

// Bit budgets of the packed words.  A top-grid entry is {slot | level << kGridSlotBits, word}: the level is < 8 (the
// grid spans at most 6 node levels), which leaves 29 bits for the slot.  A hit-list entry is {kHitValid | slot |
// (count - 1) << hit_slot_bits(SPP)}: count <= SPP, so the slot gets 31 - ceil(log2 SPP) bits -- 28 at SPP <= 8, 26 at
// SPP 32.  A tree renders through the fast / batched kernels at a given SPP while its leaf slots fit BOTH budgets,
// i.e. up to 2^28 - 1 slots = 33 M nodes at the benchmark's SPP 6.
// kHitValid: the hand-off buffer holds a pixel's sorted thresholds (non-negative floats: top bit clear) until the
// traversal overwrites the first entries with its hit list, so the first word with a clear top bit ends the list --
// no terminator is ever written (round 1 wrote one scattered dword per ray: most of the kernel's HBM writes).
constexpr int kGridSlotBits = 29;
constexpr uint32_t kGridSlotMask = (1u << kGridSlotBits) - 1u;
__host__ __device__ constexpr int hit_slot_bits(int spp) {
    return spp <= 1 ? 31 : spp <= 2 ? 30 : spp <= 4 ? 29 : spp <= 8 ? 28 : spp <= 16 ? 27 : 26;
}
__host__ __device__ inline bool nodew_is_leaf(uint32_t w) { return (w >> 30) == 2u; }

struct TreeDev {
    const uint16_t* data;   // fp16 bits [capacity*N3*data_dim]   (reference tree.data)
    const int32_t* child;   // [capacity*N3]                        (reference tree.child)
    const uint32_t* nodew;  // [capacity*8] traversal image (N == 2 only), else nullptr
    float offset[3];
    float scale[3];
    int N, N3, data_dim;
    int format, basis_dim;
    float ndc_width, ndc_height, ndc_focal;  // ndc_width <= 0: off (data_spec.hpp:49)
    int max_depth;                            // levels of child[] visited to reach the deepest leaf
    // Dense shortcut over the top of the tree (N == 2): G = top_levels bits per axis; entry
    // [(x*2^G + y)*2^G + z] = {slot | level << kGridSlotBits, nodew[slot]} where the root-path walk of that cell
    // over node levels 0..G-1 ends (at a leaf, or at the level G-1 slot).  A march step that restarts
    // above level G costs this one 8-byte, L2-resident load.  nullptr / 0 when absent.
    const uint2* topgrid;
    int top_levels;
    // Two-level traversal image (rto_abi.cpp build_wide_image; nullptr: absent).  ONE array: entries [0, 8^G) are the top-grid
    // cells, padded to wide_grid_nodes nodes of 64; wide node k is node wide_grid_nodes + k.  Entry of a node: index
    // (x2 << 4 | y2 << 2 | z2), two bits per axis; of the grid: (x << 2G | y << G | z).  A word: leaf = kLeafTag | level << kWideLevelShift |
    // sigma fp16, internal = the NODE NUMBER of the wide node below (two levels down; from the grid: the level-G node's).  An
    // entry's index is also the hit index of its leaf: wgslot[grid cell] / worig[wide node] translate it to the leaf's slot.
    const uint32_t* widew;
    const uint32_t* wgslot;
    const uint32_t* worig;
    uint32_t wide_entries;     // wide nodes * 64
    uint32_t wide_grid_nodes;  // nodes of 64 entries the grid part occupies
    // Aligned copy of the SH coefficients for shading (dense SH9 / SH16 trees, N == 2; SH25 gains nothing from it and
    // keeps data[]): per slot the 3 B coefficients in data[]'s order, zero-padded to 64 B (SH9) / 128 B (SH16) so that
    // a record is ONE 128-byte line fetched by 16-byte loads; nullptr when absent (shading then reads data[]).  When it
    // exists the host releases `data` and `child` after the upload (both nullptr until the generic kernel asks for them).
    const uint16_t* shrec;
    // Round 5: the records are laid out in the order of the TWO-LEVEL IMAGE's entries (record e = the leaf of entry e of widew;
    // a leaf at the first level of a pair owns 8 entries and so 8 copies): a hit entry of the traversal then names its record
    // directly -- no wgslot / worig / nodew look-ups between traversal and shading (they cost the shading kernel 0.37 ms per
    // 100 frames, or the traversal's flush two dependent gathers per round).  0: shrec is indexed by the leaf slot.
    int rec_by_entry;
    // RTO_TREE_COMPACT_RECORDS: shrec holds records only for the leaf slots a ray can hit (density > 0), in slot order;
    // recidx[slot] = its record, kNoRecord for the others.  nullptr: shrec is indexed by the slot itself.
    const uint32_t* recidx;
    // Quantised tree rendered WITHOUT expansion (the inputs of n3tree.cpp:279-340):
    // `data` is nullptr; per leaf slot one record of q_rec u16 values, `qrec[slot * q_rec + ...]`:
    //   [3 * q_retain] fp16 retained coefficients, (basis k, channel c) at k * 3 + c
    //   [n_basis - q_retain] codebook indices of the quantised basis functions
    //   (+ one pad value when the count is odd)
    // and per quantised basis function a 65536-entry codebook of {r, g, b, 0} fp16 (8 B entries).
    // Same bytes as quant_map + data_retained of the file, slot-major so that one hit leaf touches
    // one or two cache lines instead of one per basis function.
    // Empty-space culling (round 3): the cubes, no finer than kOccLevel, that together contain every leaf of positive
    // density, as bounding spheres in WORLD coordinates {x, y, z, radius} (radius = half diagonal + a margin far above the
    // float error of a ray's sample points).  mark_tiles_kernel projects them into each frame: an 8x8-pixel tile that no
    // sphere touches holds only rays that can never meet density -- background pixels, known without marching.
    const float4* occ_cells;
    int n_occ_cells;  // (0 with occ_cells != nullptr: a tree without density; occ_cells == nullptr: no culling)
    const uint16_t* qrec;
    const uint2* qcolors;  // [n_basis - q_retain][65536]
    int q_retain;
    int q_rec;
};

struct CamDev {
    int width, height;
    float fx, fy;
    float transform[12];
};

// The RenderOptions fields the offscreen kernel reads (render_options.hpp:13-78)
struct OptDev {
    float step_size, sigma_thresh, background_brightness;
    float render_bbox[6];
    int basis_minmax[2];
    // rodrigues(opt.rot_dirs, vdir) (volrend.cu:58-73,155): the per-frame constants of the rotation,
    // computed once on the host (make_opt_dev); rot_on = 0 below the reference's 1e-6 angle cut-off
    int rot_on;
    float rot_k[3], rot_cos, rot_sin;
    double rot_omc;  // (1.0 - cos_angle), a double in the reference's expression
};

// Strip-interleaved tile order: tile-row r belongs to XCD ((r / strip_rows) % 8); workgroup b
// runs on XCD (b % 8) (observed round-robin placement; speed only, never correctness).
struct TileMap {
    int tiles_x, tiles_y;  // 32x8-pixel tiles
    int strip_rows;        // tile rows per strip
    int per_xcd;           // workgroups per XCD (grid = 8 * per_xcd)
};

// One frame of a batched launch (render_persist): its camera, its RNG base state and where its
// pixels go.  width/height are shared by the batch.
constexpr int kMaxBatch = 128;
constexpr int kFrameChunk = 32;  // frame descriptors that travel in one kernarg (a kernarg segment holds 4 KB)
struct FrameDesc {
    float fx, fy;
    float transform[12];
    uint64_t rng_state, rng_inc;
    float* aux;
    float* image;
    uint32_t* hits;  // SPP * H*W words: thresholds, then packed hit entries (kHitValid) (traversal -> shading; layout: hit_index)
};
constexpr uint32_t kHitValid = 0x80000000u;
constexpr int kMaxQueues = 8;  // XCDs of an MI355X
// ctx queue memory (u64 words): [2..7] debug counters, [8 + 8k] next ray of queue k
constexpr int kQueueWords = 8 + 8 * kMaxQueues;
struct FrameBatch {
    int n, width, height;
    int tile_major;  // queue order: 0 = frame after frame, 1 = tile after tile (each tile for all frames in turn)
    const uint32_t* tile_order;  // [tiles8_x*tiles8_y] (ty << 16 | tx) in queue order, or nullptr
    // Ray queues of the persistent kernel: queue k holds, frame after frame, the tiles
    // tile_order[qstart[k] .. qstart[k+1]) -- one image wedge.  A wave draws from the queue of the XCD
    // it runs on (each XCD has its own L2: one wedge = one slice of the tree per L2) and steals from
    // the others once that is empty.  n_queues = 1: a single queue over whole frames.
    int n_queues;
    int qstart[kMaxQueues + 1];
    // per frame `mask_words` u32: bit t = 8x8 tile t (row-major) may hold a ray that meets density; the LAST word != 0 = keep
    // every tile of the frame (camera too close to a cell for the projection bound).  nullptr: no culling.
    const uint32_t* tile_mask;
    int mask_words;
    // The ray queues as lists (round 3): queue k = qlist[qstart[k] * n ...], qcount[k] live tile slots, each entry
    // {frame << 20 | tile y << 10 | tile x} -- the tile slots of the queue order (tile-major or frame-major over tile_order)
    // whose tile is marked, compacted in that order by the three queue_*_kernel launches before the traversal (in chunks of
    // kQueueChunk slots: qchunk[k] = first chunk of queue k, chunk_count / chunk_base = the scan's scratch).
    uint32_t* qlist;
    uint32_t* qcount;
    uint32_t* chunk_count;
    uint32_t* chunk_base;
    int qchunk[kMaxQueues + 1];
    const FrameDesc* f;  // [n] in device memory (the context's table, written on the launch stream by write_frames_kernel)
    // rto_ctx_set_lean_outputs: the shading kernel stores the noisy image as (r, g, b, alpha) and nothing else -- no aux planes;
    // 2: and nothing at all for the pixels of unmarked tiles (sparse)
    int lean;
};
struct FrameChunk {
    FrameDesc f[kFrameChunk];
};

struct FrameOut {
    float* aux;    // [8][H][W]
    float* image;  // [H][W][4]: noisy image when opt.denoise, else final (volrend.cu:206)
    unsigned long long* stats;  // nullptr, or kStatsWords counters (fast kernel, stats instantiation)
    // stats only: the tile marks of THIS frame left by a batched launch (nullptr: every tile counts as marched) -- the
    // counting kernel then also reports the work of the rays the batched path really marches (stats[6..])
    const uint32_t* stat_marks;
    int stat_mask_words;
    // single-frame culling (round 4): the tile marks of THIS frame (mark_tiles_one_kernel ran before on the stream); an 8x8 tile
    // without its bit holds only rays that meet no density: its wave writes the background and is gone.  nullptr: march all.
    const uint32_t* cull_marks;
    int cull_mask_words;
};
// stats[0..5] = SURVEY 8d's units over EVERY ray (orc_stats order: rays, rays_in_box, steps, levels of a root-restart walk,
// hit leaves, rays with a hit); stats[6..11] = the same frame as the batched path works through it: rays of marked tiles,
// their march steps, top-grid entries loaded (8 B each), traversal-image words loaded (4 B each), hit entries written,
// rays of marked tiles that entered the volume, entries of the two-level image loaded (4 B each: what render_persist loads
// instead of the traversal-image words when the tree has that image)
constexpr int kStatsWords = 14;

}  // namespace rto
