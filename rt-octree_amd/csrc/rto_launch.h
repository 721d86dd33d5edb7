// rto_launch.h -- host-callable launchers of the gfx950 kernels (defined in *.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rto_kernel_types.h"

namespace rto {

// traversal image for N == 2 trees; *bad_flag (device int, pre-zeroed) is set if an offset cannot
// be encoded
hipError_t launch_build_nodew(const int32_t* child, const uint16_t* data, int64_t n_slots, int data_dim,
                              uint32_t* nodew, int* bad_flag, hipStream_t stream);

// aligned copy of the SH coefficients (TreeDev::shrec): `rec` halves per slot = shrec_halves(basis_dim)
hipError_t launch_build_shrec(const uint16_t* data, int64_t n_slots, int data_dim, int rec, const uint32_t* recidx, uint16_t* out,
                              hipStream_t stream);
// child[] / data[] (reference layout) rebuilt from the traversal image and the aligned coefficient copy
hipError_t launch_rebuild_reference(const uint16_t* shrec, const uint32_t* nodew, const uint32_t* recidx, int64_t n_slots, int data_dim,
                                    int rec, uint16_t* data, int32_t* child, hipStream_t stream);

// entry-ordered records (TreeDev::rec_by_entry): built from data[] through the tree's two-level image (tree.widew / wgslot /
// worig / nodew must be set), and data[]'s coefficients back from them (after launch_rebuild_reference with shrec = nullptr)
hipError_t launch_build_shrec_wide(const TreeDev& tree, const uint16_t* data, int64_t n_entries, int rec, uint16_t* out, hipStream_t stream);
hipError_t launch_rebuild_reference_wide(const TreeDev& tree, int64_t n_entries, int rec, uint16_t* data, hipStream_t stream);

// top-of-tree shortcut grid: 2^(3G) entries (TreeDev::topgrid)
hipError_t launch_build_topgrid(const uint32_t* nodew, int G, uint2* grid, hipStream_t stream);

TileMap make_tile_map(int width, int height, int strip_rows);

// kernel: 1 = generic, 2 = fast.  spp must be one of {1,2,3,4,6,8,16,32} (hipErrorInvalidValue otherwise)
hipError_t launch_render(int kernel, int spp, const TreeDev& tree, const CamDev& cam, const OptDev& opt,
                         const Pcg32& rng, const PcgJumpEntry* jump, const FrameOut& fo, int strip_rows, hipStream_t stream);

// tile marks of ONE frame for the single-frame kernel's culling (FrameOut::cull_marks): zeroes `mask` ((tiles + 31) / 32 + 1
// words) on the stream and projects the tree's culling cells into the camera
hipError_t launch_mark_tiles_one(const TreeDev& tree, const CamDev& cam, uint32_t* mask, int mask_words, hipStream_t stream);

// writes n frame descriptors (host memory, read before the call returns) into a device table on `stream`
hipError_t launch_write_frames(const FrameDesc* host, int n, FrameDesc* dev_table, hipStream_t stream);

// occupancy of the persistent kernel, cached per render context (which is per device and per thread): the
// answer depends on the instantiation and on its dynamic LDS size (deeper trees need more)
struct OccupancyCache {
    const void* fn = nullptr;
    size_t lds = 0;
    int blocks_per_cu = 0;
    int cap = 0;  // tuning ("blocks_per_cu"): launch at most this many workgroups per CU (0 = what fits)
    // out: the device refused the dynamic LDS this launch needs (hipFuncSetAttribute failed); NOTHING was launched and
    // launch_render_batch returned hipSuccess -- the caller renders the frames with the generic kernel instead
    bool lds_refused = false;
    bool force_lds_refusal = false;  // test hook
};

// persistent batched renderer (N == 2 trees): fb.n frames in one launch (traversal kernel, then the
// shading kernel); `queue` = kQueueWords u64 (zeroed by queue_scan_kernel on the stream before every traversal launch); ev = nullptr or 4 events recorded before the thresholds kernel, before / after the traversal, after the shading
hipError_t launch_render_batch(int spp, const TreeDev& tree, const OptDev& opt, const FrameBatch& fb,
                               const PcgJumpEntry* jump, unsigned long long* queue, uint32_t* hits, int num_cus,
                               int refill, bool cull, OccupancyCache* occ, hipEvent_t* ev, hipStream_t stream);

// quant_map [nq][ns] + data_retained [nr][ns][3] -> slot-major records of `rec` u16 (TreeDev::qrec)
hipError_t launch_pack_quant(const uint16_t* qmap, const uint16_t* retained, int64_t ns, int nr, int nq, int rec,
                             uint16_t* out, hipStream_t stream);

hipError_t launch_rgba8(const float* rgba, uint8_t* out, int64_t n_pixels, hipStream_t stream);

#ifdef RTO_DBG_COUNTERS
hipError_t debug_shade_phases(unsigned long long* out16, bool reset);  // tools/dbg_shade_phases.py
#endif
}  // namespace rto
