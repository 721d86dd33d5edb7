/*
 * rto.h -- C ABI of the MI355X-native RT-Octree render path (librto.so).
 *
 * Drop-in boundary for the reference's operator `volrend::launch_renderer` and the objects it
 * takes (reference paths are relative to /root/reference):
 *
 *   rto_tree      <- volrend::N3Tree            renderer/include/volrend/n3tree.hpp, src/n3tree.cpp:111-362,
 *                                               src/cuda/n3tree.cu:9-49   (tree.npz -> device arrays)
 *   rto_camera    <- internal::CameraSpec       renderer/include/volrend/internal/data_spec.hpp:11-24
 *   rto_options   <- volrend::RenderOptions     renderer/include/volrend/render_options.hpp:13-78
 *   rto_ctx       <- volrend::RenderContext     renderer/include/volrend/render_context.hpp:14-214
 *   rto_launch_renderer <- launch_renderer      renderer/include/volrend/cuda/renderer_kernel.hpp:11-16,
 *                                               src/cuda/volrend.cu:236-285
 *   rto_filtering <- denoiser::filtering        denoiser/extension/filtering.h:7-13, filtering.cu:701-717
 *   rto_ctx_download_rgba8 <- the u8 conversion renderer/main_headless.cpp:521-538
 *
 * Plain pointers and sizes only; no torch / HIP types in the signatures (streams are passed as
 * `void*` = hipStream_t, NULL = the default stream).  All device pointers returned by the
 * accessors are plain linear hipMalloc memory (the reference's cudaArray/surface/texture objects
 * are replaced by linear buffers with the same logical layout).
 *
 * Threading: handles are thread-compatible, not thread-safe -- one rto_ctx (and one rto_guidance_net) per host thread
 * and per stream at a time: a context owns the hand-off buffers, ray queues and frame table of the launch in flight,
 * and two batched launches of one context on different streams would share them.  Two more places hold lazily built,
 * shared state: (1) a tree whose upload released child[] / data[] (dense SH9 / SH16, the default) rebuilds them under a lock
 * at the FIRST launch that selects the generic kernel (rto_ctx_set_kernel(RTO_KERNEL_GENERIC), N != 2, the per-frame
 * fallback of a batched launch) -- make that first launch before other threads render the same tree, or load the tree with
 * RTO_TREE_KEEP_REFERENCE; (2) the *_culled denoise entry points measure the network's background tile once per background
 * brightness, synchronising the stream and rewriting the handle's copy of it -- work queued on ANOTHER stream with the
 * previous brightness must have finished (one network handle per stream, as above, makes that automatic).
 *
 * Error convention: every function returning int returns RTO_OK (0) or a negative RTO_E_* code;
 * rto_last_error() gives the message for the calling thread.  The library never calls exit()
 * (the reference does: src/cuda/common.cu:8-21) and never falls back to a CPU path: without a
 * usable HIP device every entry point that needs one fails with RTO_E_HIP.
 */
#ifndef RTO_H
#define RTO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTO_OK 0
#define RTO_E_INVALID -1     /* bad argument */
#define RTO_E_SPP -2         /* spp not in {1,2,3,4,6,8,16,32} (volrend.cu:266-278) */
#define RTO_E_UNSUPPORTED -3 /* feature outside the headless path (probe, SG/ASG) */
#define RTO_E_HIP -4         /* HIP runtime error / no device */
#define RTO_E_IO -5          /* file missing / malformed */
#define RTO_E_FORMAT -6      /* npz content violates the tree schema (n3tree.cpp:283-291,345) */

#define RTO_AUX_CHANNELS 8   /* render_context.hpp:23 */
#define RTO_BASIS_MAX 25     /* render_options.hpp:7 */

/* data_format.hpp:8-14 */
enum { RTO_FMT_RGBA = 0, RTO_FMT_SH = 1, RTO_FMT_SG = 2, RTO_FMT_ASG = 3 };

/* RenderOptions, field for field (render_options.hpp:13-78).  bools are ints. */
typedef struct rto_options {
    float step_size;              /* 1e-4 */
    float sigma_thresh;           /* 1e-2 */
    float stop_thresh;            /* 1e-2; parsed, unused by regular tracking */
    float background_brightness;  /* 1.0 */
    float render_bbox[6];         /* {0,0,0,1,1,1} */
    int basis_minmax[2];          /* {0,24} */
    float rot_dirs[3];            /* {0,0,0} */
    int show_grid;                /* false; GUI */
    int grid_max_depth;           /* 4; GUI */
    int render_depth;             /* false; unused by the kernel */
    int enable_probe;             /* false; GUI probe -> RTO_E_UNSUPPORTED when set */
    float probe[3];               /* {0,0,1} */
    int probe_disp_size;          /* 100 */
    int denoise;                  /* true: kernel writes the noisy image, else the final image */
    int spp;                      /* 1 */
} rto_options;

/* CameraSpec (data_spec.hpp:11-24).  transform = 4x3 column-major camera-to-world
 * (camera.hpp: glm::mat4x3): [0..8] rotation columns, [9..11] centre.  Passed to the kernel by
 * value: no per-frame H2D copy (camera.cpp:67-75 in the reference). */
typedef struct rto_camera {
    int width, height;
    float fx, fy;
    float transform[12];
} rto_camera;

typedef struct rto_tree_info {
    int64_t capacity;   /* nodes */
    int N;              /* branching per axis (2) */
    int data_dim;       /* 3*basis_dim+1, or 4 for RGBA */
    int format;         /* RTO_FMT_* */
    int basis_dim;      /* -1 for RGBA */
    float scale[3];     /* invradius3 */
    float offset[3];
    int use_ndc;
    float ndc_width, ndc_height, ndc_focal;
    int max_depth;      /* deepest leaf level (levels of child[] visited to reach it) */
    int64_t device_bytes;
    int64_t wide_nodes; /* nodes of the two-level traversal image the batched kernel walks (0: it walks the one-level image) */
} rto_tree_info;

typedef struct rto_tree rto_tree; /* opaque */
typedef struct rto_ctx rto_ctx;   /* opaque */

/* Which traversal kernel rto_launch_renderer uses.  Results are bit-identical. */
enum {
    RTO_KERNEL_AUTO = 0,    /* fast path when the tree allows it (N == 2), else generic */
    RTO_KERNEL_GENERIC = 1, /* root-restart descent, one thread per pixel (any N) */
    RTO_KERNEL_FAST = 2     /* integer descent + ancestor stack (N == 2) */
};

const char* rto_version(void);
const char* rto_last_error(void);
/* number of HIP devices visible, or RTO_E_HIP */
int rto_device_count(void);

/* ---- options (render_options.hpp) ---- */
void rto_options_default(rto_options* o);
/* Parses the reference's options JSON (options/opt.json).  Like NLOHMANN_DEFINE_TYPE_INTRUSIVE
 * (render_options.hpp:61-77) all 11 keys are required; render_bbox/basis_minmax/rot_dirs keep
 * their defaults. */
int rto_options_from_json_file(const char* path, rto_options* o);
int rto_options_from_json(const char* text, rto_options* o);

/* ---- tree (N3Tree) ---- */
/* N3Tree::open (n3tree.cpp:111-154) + load_cuda (n3tree.cu:9-41).  Reads `tree.npz` (dense fp16
 * `data`, or the quantised set quant_colors/quant_map/sigma[/data_retained]) and uploads it. */
int rto_tree_load_npz(const char* path, int device, rto_tree** out);
/* flags: RTO_TREE_QUANT_DIRECT keeps a quantised tree (quant_colors / quant_map / sigma
 * [/ data_retained]) as stored and renders straight from the codebooks instead of expanding it to
 * the dense fp16 layout: same pixels, a fraction of the footprint.  N == 2,
 * SH4/9/16/25 only; such a tree renders through the batched kernels (also for single frames).
 * Ignored for dense files. */
#define RTO_TREE_QUANT_DIRECT 1
/* RTO_TREE_COMPACT: do not build the aligned copy of the SH coefficients the shading kernels otherwise read (dense SH9 /
 * SH16 trees: + 64 / 128 B per leaf slot, i.e. the device footprint roughly doubles, for ~11 % faster shading = ~2 % more
 * frames/s on the benchmark scene).  Same pixels either way. */
#define RTO_TREE_COMPACT 2
/* RTO_TREE_KEEP_REFERENCE: a dense SH9 / SH16 tree that has the aligned copy renders through the fast and the batched
 * kernels from the traversal image + that copy alone, so by default the upload RELEASES the reference-layout child[] and
 * data[] arrays once the derived ones exist (device footprint of the benchmark tree: 2.2 GB instead of 4.0 GB for a
 * 1.7 GB file) and rebuilds them -- the same leaf values -- on the first launch that selects the generic kernel
 * (rto_ctx_set_kernel(RTO_KERNEL_GENERIC)), which then pays one pass over the tree.  This flag keeps them resident. */
#define RTO_TREE_KEEP_REFERENCE 4
/* RTO_TREE_COMPACT_RECORDS: the aligned coefficient copy holds a record only for the leaf slots a ray can hit -- density
 * > 0: a hit needs sigma > sigma_thresh (rt_core.cuh:252) -- found through a 4-byte-per-slot index (8.6 M of 17 M slots of
 * the benchmark tree: 1.3 instead of 2.2 GB resident, 0.75x the file).  Same pixels; the shading kernels pay one more
 * dependent gather per hit leaf (measured: DESIGN_HISTORY.md section 3).  Launches with sigma_thresh < 0 are refused for such a
 * tree (RTO_E_UNSUPPORTED).  Ignored where no aligned copy is built (RTO_TREE_COMPACT, SH25, RGBA, quantised-direct). */
#define RTO_TREE_COMPACT_RECORDS 8
/* RTO_TREE_NO_CULLING: do not build the empty-space culling cells (the batched path then marches every ray, as rounds 1-2
 * did).  By default an N == 2 tree carries the cubes (no finer than 1/128 of the volume) that contain its leaves of
 * positive density; every batched launch projects them into its frames and skips the 8x8-pixel tiles none of them
 * touches -- rays that provably never meet density are background pixels.  Same pixels either way. */
#define RTO_TREE_NO_CULLING 16
int rto_tree_load_npz_ex(const char* path, int device, int flags, rto_tree** out);
/* Same upload from host arrays: child int32 [capacity*N^3], data fp16 bits
 * [capacity*N^3*data_dim], data_format like "SH9"/"SH16"/"RGBA" (DataFormat::parse,
 * n3tree.cpp:55-78). */
int rto_tree_from_arrays(const int32_t* child, const uint16_t* data, int64_t capacity, int N,
                         int data_dim, const char* data_format, const float scale[3],
                         const float offset[3], int device, rto_tree** out);
/* the same with the RTO_TREE_* flags of rto_tree_load_npz_ex (RTO_TREE_QUANT_DIRECT does not apply to arrays) */
int rto_tree_from_arrays_ex(const int32_t* child, const uint16_t* data, int64_t capacity, int N,
                            int data_dim, const char* data_format, const float scale[3],
                            const float offset[3], int device, int flags, rto_tree** out);
/* main_headless.cpp:400-405 (llff): switch the NDC warp on. width <= 0 turns it off. */
int rto_tree_set_ndc(rto_tree* t, float ndc_width, float ndc_height, float ndc_focal);
int rto_tree_get_info(const rto_tree* t, rto_tree_info* info);
/* Host-only: runs the same npz parsing + N3Tree::load_npz decode as rto_tree_load_npz (no device
 * needed) and writes a JSON description into json_out: schema fields plus FNV-1a-64 checksums of
 * the child[] and (decoded) data[] arrays.  Returns RTO_E_INVALID if `cap` is too small. */
int rto_tree_probe_npz(const char* path, char* json_out, size_t cap);
void rto_tree_free(rto_tree* t);

/* ---- render context (RenderContext) ---- */
/* RenderContext::update (render_context.hpp:70-91): aux [8][H][W] f32, noisy and final images
 * [H][W][4] f32, rng = pcg32(20230418) (:16).  offscreen is always true (headless path). */
int rto_ctx_create(int width, int height, int device, rto_ctx** out);
/* The same with `frames` (1..128) frame slots: aux [frames][8][H][W], noisy / image [frames][H][W][4],
 * contiguous.  Single-frame entry points and accessors act on the slot chosen with
 * rto_ctx_select_frame (default 0); rto_launch_renderer_batch fills slots 0..n-1. */
int rto_ctx_create_batch(int width, int height, int frames, int device, rto_ctx** out);
int rto_ctx_frames(const rto_ctx* c);
int rto_ctx_select_frame(rto_ctx* c, int frame);
int rto_ctx_selected_frame(const rto_ctx* c);
void rto_ctx_free(rto_ctx* c);
int rto_ctx_width(const rto_ctx* c);
int rto_ctx_height(const rto_ctx* c);
float* rto_ctx_aux(rto_ctx* c);    /* device, planar [8][H][W]: r,g,b,a,r^2,g^2,b^2,a^2 */
float* rto_ctx_noisy(rto_ctx* c);  /* device, [H][W][4] (reference: noisy_image_arr) */
float* rto_ctx_image(rto_ctx* c);  /* device, [H][W][4] (reference: image_arr / surf_obj) */
/* ctx.rng (pcg32): seed(initstate, initseq) pcg32.h:53-59; advance(delta) :145-166 (the
 * reference's per-frame `ctx.rng.advance()` is delta = 1<<32) */
void rto_ctx_rng_seed(rto_ctx* c, uint64_t initstate, uint64_t initseq);
void rto_ctx_rng_advance(rto_ctx* c, int64_t delta);
void rto_ctx_rng_set(rto_ctx* c, uint64_t state, uint64_t inc);
void rto_ctx_rng_get(const rto_ctx* c, uint64_t* state, uint64_t* inc);
/* choose the traversal kernel (RTO_KERNEL_*); default AUTO */
int rto_ctx_set_kernel(rto_ctx* c, int kernel);
/* Performance knobs; never change results.  key: "strip_rows" (single-frame kernel: tile rows per XCD
 * strip, >= 1); batched kernel: "refill" (0 = default; 100 * waves/SIMD + idle-lane threshold selects one of the A/B instantiations), "tile_order"
 * (0 = row-major tiles, 1 = centre-out), "xcd_queues" (1 = one ray queue per XCD over its share of the image,
 * with stealing; 0 = a single queue), "tile_major" / "tile_block" (queue order), "queue_bands" (the XCD queues take bands of this many
 * 8-pixel tile rows, band j -> queue j % 8; 0 = angular wedges around the image centre; default 3), "cull" (1 = skip the tiles whose rays provably meet no density, the default; 0 = march every ray), "blocks_per_cu" (0 = as many
 * workgroups of the persistent traversal kernel per CU as fit, else a cap 1..8: the kernel's true occupancy knob --
 * `refill`'s waves/SIMD only sets the register budget). */
int rto_ctx_set_tuning(rto_ctx* c, const char* key, int value);
/* Lean outputs of the batched render -> denoise route (round 5; off by default).  The reference's kernel stores, per pixel,
 * 8 aux planes and an RGBA32F image (volrend.cu:187-212: 48 bytes) -- of which its own denoise stage reads aux planes 0..3
 * (planes 4..7 are their squares) and the image's rgb, which duplicates planes 0..2.  With lean outputs on, a
 * rto_launch_renderer_batch with options.denoise = 1 stores 16 bytes per pixel instead: the noisy image as (r, g, b, ALPHA)
 * -- alpha = aux plane 3 where the reference writes 1.0 -- and NO aux planes (the context's aux buffer keeps whatever it
 * held).  Consumers: rto_denoise (picks the route by itself), or rto_guidance_net_forward*(..., flags = RTO_NET_INPUT_RGBA)
 * on rto_ctx_noisy + rto_filtering* on the same image (the filter never reads the image's alpha, filtering.cu:186-199).
 * The denoised image is bit-identical to the full-output route's.  Single-frame launches, launches with denoise = 0 and
 * everything that reads the aux buffer (--write_buffer, rto_ctx_download_aux) need the full outputs: leave it off there.
 * The state is kept PER FRAME SLOT: a launch changes it only for the slots it writes (a single-frame launch into slot k of a
 * lean batch leaves the other slots lean).  rto_ctx_frames_are_lean: 1 when slots [first_slot, first_slot + n) were all
 * written by a lean launch last, 0 when none was, -1 for a mixed range -- which rto_denoise refuses (RTO_E_INVALID): denoise
 * each run of slots by itself. */
int rto_ctx_set_lean_outputs(rto_ctx* c, int level);
int rto_ctx_frames_are_lean(const rto_ctx* c, int first_slot, int n);
/* Level 2 (round 6), SPARSE lean outputs: as level 1, and the launch stores NOTHING for the pixels of the 8x8 tiles its
 * empty-space culling left unmarked (two thirds of the bench scene's pixels; they are the background: colour =
 * background_brightness, alpha 0).  The noisy image is then valid inside marked tiles only; its consumers need the launch's
 * tile marks (rto_ctx_tile_marks): rto_denoise with RTO_FILTER_FACTORISED does it by itself, the two-call form passes
 * RTO_NET_INPUT_RGBA | RTO_NET_INPUT_SPARSE and the marks to rto_guidance_net_forward_packed_culled + rto_filtering_packed_culled.
 * rto_ctx_download_image / _rgba8 of the noisy image fill the unmarked tiles in on the host.  The denoised image is complete and
 * bit-identical to the other levels'.  rto_ctx_frames_lean_level: 0 / 1 / 2 when all of the slots agree, -1 otherwise. */
int rto_ctx_frames_lean_level(const rto_ctx* c, int first_slot, int n);
/* Per-kernel HIP-event timing of the batched path: when enabled, every rto_launch_renderer_batch
 * records events before the traversal kernel, between it and the shading kernel, and after (on the
 * launch stream; up to 256 launches between reads).  _read synchronises on the recorded events and
 * returns the mean milliseconds per launch of each kernel, then resets the ring. */
/* Diagnostic (synchronises the device): of the last rto_launch_renderer_batch on this context, how many 8x8-pixel tile
 * slots (tile x frame) were marched and how many there were -- the rest were culled as provably empty (see
 * RTO_TREE_NO_CULLING; tuning key "cull" = 0 switches the culling off per context). */
int rto_ctx_queue_stats(rto_ctx* c, int64_t* live_tile_slots, int64_t* all_tile_slots);
/* The tile marks the last rto_launch_renderer_batch left on the device: frames x words_per_frame uint32 (bit t of a frame's
 * words = 8x8 tile t, row-major, may hold a ray that meets density; bit 0 of the frame's last word = treat every tile as
 * marked).  An unmarked tile's pixels are exactly the background: colour = *background, alpha 0.  Valid (stream-ordered
 * after that launch) until the next launch on the context; RTO_E_INVALID when the last launch was a single-frame one.
 * Frame f of the marks is context slot first_slot + f.  For rto_filtering_packed_culled. */
int rto_ctx_tile_marks(const rto_ctx* c, const uint32_t** marks, int* words_per_frame, int* first_slot, int* frames, float* background);
int rto_ctx_kernel_timing(rto_ctx* c, int enable);
int rto_ctx_kernel_timing_read(rto_ctx* c, float* traverse_ms, float* shade_ms, int* launches);
/* the same with the thresholds kernel (sample_kernel: RNG jump, SPP draws, sort for every pixel of the batch) reported too,
 * as raygen_ms */
int rto_ctx_kernel_timing_read3(rto_ctx* c, float* raygen_ms, float* traverse_ms, float* shade_ms, int* launches);
/* Work counters for the roofline's ALGORITHMIC byte count.  When enabled, the fast
 * kernel's counting instantiation runs instead of the timed one and accumulates, over the launches
 * since the last rto_ctx_get_stats(reset=1): {rays, rays_in_box, march steps, descent levels a
 * root-restart walk visits, distinct hit leaves, rays with a hit}.  Never enable it in a timed run. */
int rto_ctx_enable_stats(rto_ctx* c, int enable);
int rto_ctx_get_stats(rto_ctx* c, void* stream, uint64_t out[6], int reset);
/* enable = 2: additionally count the frame as the BATCHED path works through it (round 4: the figure above prices a
 * root-restart walk of every ray, rt_core.cuh:241-270 + n3tree_query.hpp:22-47, which the batched kernels do not perform).
 * The caller renders frames with rto_launch_renderer_batch, selects a slot and re-renders that slot's pose (same RNG) with
 * rto_launch_renderer: rays of tiles the batched launch culled are left out, and a node visit counts as the one load
 * render_persist issues for it.  out = {rays of marked tiles, their march steps, top-grid entries loaded (8 B each),
 * traversal-image words loaded (4 B each), hit entries written (4 B each), rays of marked tiles that entered the volume,
 * entries of the two-level traversal image loaded (4 B each: render_persist loads these INSTEAD of the traversal-image words
 * when the tree has that image, rto_tree_info.wide_image), 0}. */
int rto_ctx_get_march_stats(rto_ctx* c, void* stream, uint64_t out[8], int reset);

/* Diagnostics, host only (no device): the two-level traversal image the batched kernel walks (a node of level G + 2p merged
 * with its eight children: one load per two levels; derived from child[], n3tree.hpp / n3tree_query.hpp:22-47 is what it
 * must answer like) built for a breadth-first child[] and walked for n points (24-bit fixed-point x, y, z each) exactly as
 * the kernel walks it; out: the leaf's level, its slot in child[] / data[], its sigma bits, the number of wide nodes. */
int rto_wide_image_probe(const int32_t* child, const uint16_t* sigma_bits, int64_t capacity, int max_depth, int top_levels,
                         const uint32_t* points, int64_t n, int32_t* out_level, int64_t* out_slot, uint16_t* out_sigma,
                         int64_t* out_wide_nodes);

/* ---- the operator ---- */
/* launch_renderer(tree, cam, options, ctx, stream, offscreen=true) (volrend.cu:236-285).
 * Asynchronous on `stream`.  Writes ctx aux + (options->denoise ? noisy : image).
 * cam->width/height must equal the ctx size. */
int rto_launch_renderer(const rto_tree* tree, const rto_camera* cam, const rto_options* options,
                        rto_ctx* ctx, void* stream);

/* Throughput form of the operator: n frames (poses of the same tree) in ONE launch of the
 * persistent ray-queue kernel (n <= 128).  Frame f is rendered with cams[f] into frame slot f with the RNG
 * ctx.rng advanced by rng_jumps[f] * 2^32 (NULL: f jumps) -- bit-identical to n sequential
 * rto_launch_renderer calls separated by rto_ctx_rng_advance(ctx, 1<<32), the reference's frame
 * loop (main_headless.cpp:485-506).  ctx.rng itself is not modified.  Needs an N == 2 tree. */
int rto_launch_renderer_batch(const rto_tree* tree, const rto_camera* cams, const int64_t* rng_jumps, int n,
                              const rto_options* options, rto_ctx* ctx, void* stream);

/* denoiser::filtering(stream, weight_map[L,H,W], guidance_map[L,H,W], img_in, img_out)
 * (filtering.cu:701-717).  All pointers are device pointers; img_in/img_out are [H][W][4] f32
 * (the reference passes ctx.noisy_tex_obj / ctx.surf_obj, denoiser.cpp:56-57).  L in 1..6. */
int rto_filtering(void* stream, const float* weight_map, const float* guidance_map, int L, int H,
                  int W, const float* img_in, float* img_out);
/* n images per launch: weight_map / guidance_map [n][L][H][W], img_in / img_out [n][H][W][4] */
int rto_filtering_batch(void* stream, const float* weight_map, const float* guidance_map, int L, int H,
                        int W, int n, const float* img_in, float* img_out);
/* The same filter in one of two arithmetic forms.  RTO_FILTER_EXACT = rto_filtering_batch: every tap's
 * exp(g(q) - max_p) evaluated as the reference does, bit-identical to the CPU oracle.  RTO_FILTER_FACTORISED:
 * exp(g(q) - c) once per pixel with a per-tile constant c (the factor cancels in sum k rgb / sum k) and the
 * window sums as box filters -- 4 exps per pixel instead of 164 at L = 4; results agree with the exact form
 * to ~1e-6 relative (> 120 dB), inside the 1e-4 dB tolerance of the float paths, not bit for bit. */
enum { RTO_FILTER_EXACT = 0, RTO_FILTER_FACTORISED = 1 };
int rto_filtering_batch_mode(void* stream, const float* weight_map, const float* guidance_map, int L, int H,
                             int W, int n, const float* img_in, float* img_out, int mode);
/* Training side -- Filtering::forward with requires_grad and Filtering::backward
 * (filtering.cu:596-707; grad_weight_accumulate :230-248, grad_guidance_accumulate :250-301), what
 * `_denoiser.filtering_autograd` (bindings.cpp) runs under autograd.  All device pointers, fp32:
 *   forward : as rto_filtering_batch, also writing rgb_filtered [n][L][H][W][4] (alpha 0), max_map and
 *             inv_kernel_sum [n][L][H][W] (the reference keeps one [B,H,W,*] tensor per level);
 *   backward: grad_output, img_in [n][H][W][4] + the forward's inputs and saves ->
 *             grad_weight, grad_guidance [n][L][H][W] (fully overwritten).
 * The guidance gradient is gathered per pixel in a fixed order (no atomics): results are run-to-run
 * identical, which the reference's atomicAdd scatter is not. */
int rto_filtering_train_forward(void* stream, const float* weight_map, const float* guidance_map, int L, int H,
                                int W, int n, const float* img_in, float* img_out, float* rgb_filtered,
                                float* max_map, float* inv_kernel_sum);
int rto_filtering_backward(void* stream, const float* grad_output, const float* img_in, const float* weight_map,
                           const float* guidance_map, const float* rgb_filtered, const float* max_map,
                           const float* inv_kernel_sum, int L, int H, int W, int n, float* grad_weight,
                           float* grad_guidance);
/* convenience: filtering from ctx noisy -> ctx image */
int rto_ctx_filtering(rto_ctx* c, void* stream, const float* weight_map, const float* guidance_map, int L);

/* ---- outputs (main_headless.cpp:508-540) ---- */
/* final image -> RGBA8 on the device ((uint8_t)(f*255), truncation) -> host [H][W][4];
 * synchronises `stream`.  which: 0 = final image, 1 = noisy image */
int rto_ctx_download_rgba8(rto_ctx* c, void* stream, int which, uint8_t* host_out);
int rto_ctx_download_image(rto_ctx* c, void* stream, int which, float* host_out); /* [H][W][4] f32 */
int rto_ctx_download_aux(rto_ctx* c, void* stream, float* host_out);              /* [8][H][W] f32 */

/* ---- event timer (RenderContext::Timer, render_context.hpp:122-213) ---- */
enum { RTO_T_RENDER = 0, RTO_T_TORCH = 1, RTO_T_FILTER = 2 };
int rto_timer_reset(rto_ctx* c, void* stream);
int rto_timer_start(rto_ctx* c, int which);
int rto_timer_stop(rto_ctx* c, int which);
/* Timer::record(denoise): synchronise on the last stop event and accumulate */
int rto_timer_record(rto_ctx* c, int denoise);
/* Timer::report: mean ms per bucket and FPS = 1000/(render+torch+filter) */
int rto_timer_report(const rto_ctx* c, float ms_out[3], float* fps_out, int* frames_out);

/* ---- GuidanceNet forward as one fused kernel ---- */
/* The compact network of denoiser/network.py:156-168 (what compact_and_compile exports and
 * Denoiser::denoise runs, denoiser.cpp:46): relu6(conv3x3(8 -> c1)) -> relu6(conv3x3(c1 -> 2*levels))
 * -> softmax over the first `levels` channels.  Weights in PyTorch layout, fp32:
 * w1 [c1][8][3][3], b1 [c1], w2 [2*levels][c1][3][3], b2 [2*levels]; they are rounded to fp16 like the
 * reference's `.half()` module.  Supported: c1 = 32, levels = 4 (denoiser/configs/blender.txt:21-25);
 * anything else returns RTO_E_UNSUPPORTED and the caller keeps using the TorchScript module. */
typedef struct rto_guidance_net rto_guidance_net;
int rto_guidance_net_create(const float* w1, const float* b1, const float* w2, const float* b2, int c1, int levels,
                            int device, rto_guidance_net** out);
/* aux: device [n][8][H][W] fp32; outputs: device [n][levels][H][W] fp32 (weight_map, guidance_map) */
int rto_guidance_net_forward(const rto_guidance_net* net, void* stream, const float* aux, int n, int H, int W,
                             float* weight_map, float* guidance_map);
/* flags: RTO_NET_AUX_SQUARES_IMPLIED -- the caller guarantees that aux planes 4..7 are the fp32 squares of planes
 * 0..3, which is how the renderer fills them (volrend.cu:195-202); the kernel then reads planes 0..3 only and
 * squares them itself: the same values from half the bytes.  Results are bit-identical to flags = 0 on such input. */
#define RTO_NET_AUX_SQUARES_IMPLIED 1
/* RTO_NET_INPUT_RGBA (round 5) -- `aux` is not an aux buffer but an interleaved image, device [n][H][W][4] fp32 = (r, g, b,
 * alpha): the values of aux planes 0..3 (volrend.cu:187-194), as a LEAN batched launch leaves them in the context's noisy
 * buffer (rto_ctx_set_lean_outputs).  Squares implied.  Bit-identical maps to the aux-buffer input. */
#define RTO_NET_INPUT_RGBA 2
/* RTO_NET_INPUT_SPARSE (round 6; with RTO_NET_INPUT_RGBA, tile marks and the packed route only) -- the image is that of a SPARSE
 * lean launch (rto_ctx_set_lean_outputs level 2): it holds nothing for the pixels of unmarked (culled) render tiles, which are
 * the background by construction; the network takes (background x 3, alpha 0) for them and stores no maps for the tiles it
 * skips -- rto_filtering_packed_culled with the same marks substitutes both.  Same denoised images, bit for bit. */
#define RTO_NET_INPUT_SPARSE 4
int rto_guidance_net_forward_ex(const rto_guidance_net* net, void* stream, const float* aux, int n, int H, int W,
                                float* weight_map, float* guidance_map, int flags);
/* The denoise stage (Denoiser::denoise, denoiser.cpp:31-61) as two launches that keep the maps in their native
 * precision: _forward_packed runs the network and leaves its 8 output channels as fp16 (4 softmax logits + 4
 * guidance values per pixel, [n][H][W][8], in a scratch buffer the handle owns) -- the reference's `.float()` only
 * widens those values; rto_filtering_packed then runs the factorised filter (RTO_FILTER_FACTORISED) on them, taking
 * the softmax itself: img_in / img_out [n][H][W][4] fp32.  Output = rto_guidance_net_forward_ex +
 * rto_filtering_batch_mode(FACTORISED) bit for bit, from half the map bytes.  Same stream for both calls; one
 * handle per stream at a time.
 * rto_filtering_packed is told the extent of the images it is handed and refuses (RTO_E_INVALID) one that differs from
 * the maps' -- a handle shared between contexts of different batch sizes cannot overrun the smaller one.  The scratch
 * grows on demand, which synchronises the device once; rto_guidance_net_reserve sizes it up front so that a timed or
 * captured region never does. */
int rto_guidance_net_forward_packed(rto_guidance_net* net, void* stream, const float* aux, int n, int H, int W, int flags);
int rto_guidance_net_reserve(rto_guidance_net* net, int n, int H, int W);
int rto_filtering_packed(const rto_guidance_net* net, void* stream, const float* img_in, float* img_out, int n, int H, int W);
/* rto_filtering_packed that does not filter where there is nothing to filter: a 32x32 output tile whose inputs (its 40x40
 * staged pixels and the 5x5 aux neighbourhood behind each of their map values) all lie inside the frame and in unmarked tiles
 * of `tile_marks` (rto_ctx_tile_marks of the launch that rendered img_in and the network's aux; frame f of the marks = image
 * f) reads only background pixels and the network's background maps.  Every such tile computes the same 32x32 values; the
 * handle measures them once per `background` by running the two kernels on a synthetic background frame (the first call
 * with a new brightness synchronises `stream`) and copies them instead.  Output = rto_filtering_packed bit for bit.
 * tile_marks == NULL: rto_filtering_packed.
 * rto_guidance_net_forward_packed_culled is the same idea one stage earlier: a 32x8 tile of the network whose 36x12 input
 * pixels all lie inside the frame and in unmarked tiles is filled with the network's background output (8 fp16 values,
 * measured on the same synthetic frame) instead of being computed.  Maps = rto_guidance_net_forward_packed's bit for bit. */
/* The two stages on fp32 weight / guidance planes (the reference's tensors; RTO_FILTER_EXACT = the bit-exact route) with the
 * same tile skipping: the maps must be this network's output for the aux of the launch the marks belong to.  Outputs =
 * rto_guidance_net_forward_ex / rto_filtering_batch_mode bit for bit.  tile_marks == NULL: those functions. */
int rto_guidance_net_forward_culled(rto_guidance_net* net, void* stream, const float* aux, int n, int H, int W, float* weight_map,
                                    float* guidance_map, int flags, const uint32_t* tile_marks, int words_per_frame, float background);
int rto_filtering_culled(rto_guidance_net* net, void* stream, const float* weight_map, const float* guidance_map, int H, int W, int n,
                         const float* img_in, float* img_out, int mode, const uint32_t* tile_marks, int words_per_frame, float background);
int rto_guidance_net_forward_packed_culled(rto_guidance_net* net, void* stream, const float* aux, int n, int H, int W, int flags,
                                           const uint32_t* tile_marks, int words_per_frame, float background);
int rto_filtering_packed_culled(rto_guidance_net* net, void* stream, const float* img_in, float* img_out, int n, int H, int W,
                                const uint32_t* tile_marks, int words_per_frame, float background);
/* Denoiser::denoise (denoiser.cpp:31-61) in one call, for the n frames of `ctx` from its selected slot on: the network on the
 * context's aux buffer, the filter from its noisy buffer into its image buffer.  mode RTO_FILTER_FACTORISED: packed fp16
 * maps + factorised filter (the throughput route); RTO_FILTER_EXACT: fp32 planes (a scratch the handle owns, 32 B per
 * pixel) + the bit-exact filter.  When the frames are those of the last rto_launch_renderer_batch on `ctx` its tile marks
 * are used (the *_culled calls above); after a single-frame launch the plain kernels run.  Same results either way.
 * Like the packed scratch, the plane scratch grows on demand and growing synchronises the device once: call it once at the
 * largest extent before a timed or captured region. */
int rto_denoise(rto_guidance_net* net, rto_ctx* ctx, int n, int mode, void* stream);
void rto_guidance_net_free(rto_guidance_net* net);

/* ---- profiling aid ---- */
/* Counter calibration (MI355X_MICROARCH.md "HBM"): launches `repeats` kernels in which every lane
 * loads one dword from its own never-repeated 128-byte line of a zero-filled n_lines*128-byte buffer
 * (the render path's access shape).  Known lines per launch = n_lines; compare with FETCH_SIZE. */
int rto_probe_gather(uint64_t n_lines, int repeats);
/* Ceiling probes for the traversal kernel (DESIGN.md "What bounds the traversal"; tools/probe_ceiling.py).
 * rto_probe_gather_sweep: persistent waves (`wps` per SIMD on every CU) issue `iters` wave-level dword
 * gathers each; every gather touches exactly `lines_per_gather` (1..64) distinct 64-byte lines of a
 * zero-filled table of `table_bytes` (rounded down to a power of two), the lanes dealt over the lines
 * interleaved (blocked = 0) or in runs (blocked = 1); dependent = 1 chains each gather's address on the
 * previous one's value (the traversal's shape), 0 keeps four in flight per wave.
 * out = {wall ms per launch, mean shader-clock cycles per wave, waves, gathers per wave}.
 * rto_probe_valu: persistent waves (`wps` per SIMD on every CU) run `iters` times ONE asm block of exactly 32
 * wave-level VALU instructions on independent registers; `kind` picks the opcode (rto_probe_valu_name(kind) names
 * it, NULL past the last kind).  out = {wall ms, mean s_memtime ticks per wave, waves, VALU instructions per wave
 * (exact), ticks from the first wave's start to the last wave's end, CUs}. */
int rto_probe_gather_sweep(uint64_t table_bytes, int lines_per_gather, int blocked, int dependent, int wps,
                           int iters, int repeats, double out[4]);
int rto_probe_valu(int kind, int wps, int iters, double out[6]);
const char* rto_probe_valu_name(int kind);
/* rto_probe_scratch: one launch (null stream, asynchronous) of a kernel with a private segment (kind bit 0: a dynamically
 * indexed per-thread array), 34 KB of static LDS (bit 1), an MFMA loop (bit 2) -- the would-be trigger in
 * tools/contention_determinism.py's experiment on determinism when processes share the GPU. */
int rto_probe_scratch(int kind, int blocks, int iters);
/* Test hook: host_out[i] = the threshold the renderer draws from the RNG float k / 2^23, k = first_k + i
 * (rt_core.cuh:67-88 `-logf(1 - rng.next_float())` in the library's deterministic arithmetic), computed on the
 * device by the very function the kernels call.  first_k + count <= 2^23: the whole domain can be compared with
 * the oracle value by value. */
int rto_probe_thresholds(uint32_t first_k, uint32_t count, float* host_out);
/* Test hook: host_out[i] = f(x_i) computed on the device, x_i = the float with bit pattern first_bits + i * stride
 * (wrapping), f = the library's deterministic logf (fn 0: thresholds), expf (fn 1: the SH colour's sigmoid) or the
 * filter's fp32 exp (fn 2) -- the functions DESIGN.md "Math" defines in place of `__logf` / `__expf`
 * (rt_core.cuh:74,95,314; filtering.cu:191).  Lets a test sweep the whole float range against the oracle. */
int rto_probe_math(int fn, uint32_t first_bits, uint32_t stride, uint32_t count, float* host_out);
/* Test hook (round 6): the shading kernels evaluate `cnt / (1.f + expf(-t))` (rt_core.cuh:314-318) through a short form --
 * the library's expf for |t| <= 87 in fused multiply-adds, the division as reciprocal + one exact-residual correction --
 * that must return the float of the plain statement for every argument.  mode 0: the short sigmoid against the plain one
 * for the floats t with bit patterns first_bits + i, i < count (<= 2^32), times every sample count cnt_lo .. cnt_hi
 * (1 .. 32); mode 1: the short division cnt / d alone, for d = those floats (callers pass [1, 2^126)).  Both sides run on
 * the device.  out3[0] = pairs that differ, out3[1] = the first of them (bits << 8 | cnt; ~0 if none), out3[2] = pairs compared.
 * Modes 2 / 3: the shading kernels' one-instruction `(float)half * b` (v_fma_mix_f32 with a -0.0 addend; rt_core.cuh:286-312's
 * 48 products per hit entry) against convert-then-multiply, for the floats b = first_bits + i (mode 2; times the 16 special
 * halves selected by cnt 1 .. 16) or first_bits + 4099 i (mode 3; times halves 2048 (cnt - 1) .. 2048 cnt - 1: all 65536 over
 * cnt 1 .. 32), in both packed positions. */
int rto_probe_sigmoid(int mode, uint32_t first_bits, uint64_t count, int cnt_lo, int cnt_hi, uint64_t* out3);

#ifdef __cplusplus
}
#endif
#endif /* RTO_H */
