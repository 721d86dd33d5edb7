/*
 * rto_oracle.h -- CPU restatement of RT-Octree's render hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the
 * product path (rt-octree_amd/, include/).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it, and only as the checker / the reported CPU baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - pcg32:        PINNED against the reference's own header (renderer/3rdparty/pcg32.h compiled by
 *                   oracle/ref_kat/Makefile; vectors in tests/golden/pcg32_kat.json).
 *   - sample_dst:   PINNED against the SURVEY.md section 8c G2 known-answer vector (libm math mode).
 *   - npz reading:  PINNED against the reference's vendored cnpy (oracle/ref_kat, tests/golden/npz_*).
 *   - GuidanceNet:  PINNED against the imported reference module (tests/golden/guidance_*.npz).
 *   - trace_ray / render_kernel / filter:  PARITY UNPINNED by execution.  The reference's
 *     rt_core.cuh / volrend.cu / filtering.cu need cuda_runtime.h, cuda_fp16.h, a cmake-generated
 *     volrend/common.hpp and libtorch: unbuildable in this image without stand-ins, which the
 *     build rules forbid.  These functions are literal restatements, each citing file:line.
 *
 * Math modes: the reference calls the NVIDIA approximations __logf/__expf (rt_core.cuh:74,95,314;
 * filtering.cu:191) whose bit patterns cannot be reproduced off NVIDIA hardware.  The oracle
 * therefore offers
 *   ORC_MATH_DET  (0): logf/expf restated from IEEE-754 double +,-,*,/ only (orc_det_logf/expf
 *                      below) -- the definition the HIP kernels share, bit for bit;
 *   ORC_MATH_LIBM (1): glibc logf/expf, used for the SURVEY G2 vector and the PSNR-equivalence test.
 */
#ifndef RTO_ORACLE_H
#define RTO_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MATH_DET 0
#define ORC_MATH_LIBM 1
#define ORC_BASIS_MAX 25 /* render_options.hpp:7 VOLREND_GLOBAL_BASIS_MAX */
#define ORC_MAX_SPP 32

/* data_format.hpp:8-14 */
enum { ORC_FMT_RGBA = 0, ORC_FMT_SH = 1, ORC_FMT_SG = 2, ORC_FMT_ASG = 3 };

/* pcg32.h:39-201 */
typedef struct { uint64_t state, inc; } orc_pcg32;

/* internal/data_spec.hpp:25-52 (TreeSpec), host pointers */
typedef struct {
    const uint16_t* data;  /* fp16 bits [capacity*N^3*data_dim] */
    const int32_t* child;  /* [capacity*N^3] relative node offsets, 0 = leaf */
    float offset[3];
    float scale[3];
    int N;
    int data_dim;
    int format;     /* ORC_FMT_* */
    int basis_dim;  /* -1 for RGBA */
    float ndc_width; /* <=0: NDC off (data_spec.hpp:49) */
    float ndc_height;
    float ndc_focal;
} orc_tree;

/* render_options.hpp:13-78, the fields the offscreen kernel reads */
typedef struct {
    float step_size;
    float sigma_thresh;
    float stop_thresh;            /* parsed, unused by regular tracking */
    float background_brightness;
    float render_bbox[6];
    int basis_minmax[2];
    float rot_dirs[3];
    int denoise;
    int spp;
} orc_options;

/* internal/data_spec.hpp:11-24 (CameraSpec); transform = 4x3 column-major c2w */
typedef struct {
    int width, height;
    float fx, fy;
    float transform[12];
} orc_camera;

/* Counters behind SURVEY 8(d)'s ALGORITHMIC byte formula. */
typedef struct {
    uint64_t rays;          /* pixels traced */
    uint64_t rays_in_box;   /* passed the slab test */
    uint64_t steps;         /* march-loop iterations */
    uint64_t levels;        /* child[] loads (descent levels visited) */
    uint64_t hit_leaves;    /* distinct hit leaves shaded */
    uint64_t hit_rays;      /* rays with >=1 hit */
} orc_stats;

void orc_options_default(orc_options* o);
void orc_set_math_mode(int mode);
int orc_get_math_mode(void);

/* pcg32 */
void orc_pcg32_seed(orc_pcg32* r, uint64_t initstate, uint64_t initseq);
uint32_t orc_pcg32_next_uint(orc_pcg32* r);
float orc_pcg32_next_float(orc_pcg32* r);
void orc_pcg32_advance(orc_pcg32* r, int64_t delta);

/* math */
float orc_det_logf(float x);
float orc_det_expf(float x);
float orc_fexp(float x); /* fp32-only exp used by the filter */
float orc_half2float(uint16_t h);

/* pieces (exposed for known-answer tests) */
void orc_sample_dst(int spp, orc_pcg32* rng, float* dst /*[spp+1]*/);
void orc_thresholds(uint32_t first_k, uint32_t count, float* out);
void orc_math_sweep(int fn, uint32_t first_bits, uint32_t stride, uint32_t count, float* out);
/* returns leaf slot index (sub_ptr); xyz becomes leaf-local; *cube_sz = N^depth */
int64_t orc_query(const orc_tree* t, float xyz[3], float* cube_sz, int* levels);
void orc_sh_basis(int basis_dim, const float dir[3], float out[ORC_BASIS_MAX]);
int orc_trace_ray(const orc_tree* t, float dir[3], const float vdir[3], const float cen[3],
                  const orc_options* opt, float tmax_bg, float out[4], orc_pcg32* rng,
                  orc_stats* st);

/* whole frame: aux [8][H][W] fp32, rgba [H][W][4] fp32 (alpha forced to 1). returns 0 / <0 */
int orc_render_frame(const orc_tree* t, const orc_camera* cam, const orc_options* opt,
                     const orc_pcg32* rng_base, float* aux, float* rgba, orc_stats* st,
                     int num_threads);
/* one pixel (same contract), for spot checks */
int orc_render_pixel(const orc_tree* t, const orc_camera* cam, const orc_options* opt,
                     const orc_pcg32* rng_base, int idx, float out_aux8[8], float out_rgba[4],
                     orc_stats* st);

/* denoiser/extension/filtering.cu:108-228,440-470: L-level guided softmax filter.
 * weight,guidance [L][H][W]; noisy,out [H][W][4] */
int orc_filter(int L, int H, int W, const float* weight, const float* guidance,
               const float* noisy, float* out, int num_threads);
/* training side (filtering.cu:230-301,596-707): forward that also saves rgb_filtered [L][H][W][4],
 * max_map and inv_kernel_sum [L][H][W]; backward -> grad_weight, grad_guidance [L][H][W] */
int orc_filter_train_forward(int L, int H, int W, const float* weight, const float* guidance, const float* noisy,
                             float* out, float* rgb_filtered, float* max_map, float* inv_kernel_sum, int num_threads);
int orc_filter_backward(int L, int H, int W, const float* grad_out, const float* img_in, const float* weight,
                        const float* guidance, const float* rgb_filtered, const float* max_map,
                        const float* inv_kernel_sum, float* grad_weight, float* grad_guidance, int num_threads);

/* instrumentation: march steps per pixel */
int orc_frame_steps(const orc_tree* t, const orc_camera* cam, const orc_options* opt,
                    const orc_pcg32* rng_base, uint32_t* steps_out, int num_threads);

/* main_headless.cpp:535-538 */
void orc_rgba8(const float* rgba, uint8_t* out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
