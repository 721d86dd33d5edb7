// render_kernels.hip -- batched-regular-tracking octree renderer for gfx950 (MI355X).
//
// What the reference computes (relative to /root/reference):
//   render_kernel<SPP>      renderer/src/cuda/volrend.cu:84-213   pixel -> ray, RNG jump, trace, bg
//                                                                  composite, aux + image writes
//   trace_ray<float,SPP>    renderer/include/volrend/cuda/rt_core.cuh:195-332
//   query_single_from_root  renderer/include/volrend/internal/n3tree_query.hpp:13-48
//   maybe_precalc_basis     renderer/include/volrend/internal/lumisphere.hpp:38-80 (SH)
//   sample_dst<SPP>         rt_core.cuh:67-193
//
// Three traversal kernels with bit-identical results (tests/test_render_parity.py):
//   render_generic<SPP>  any N, root-restart float descent -- the plain statement of the algorithm.
//   render_fast<SPP>     N == 2, one frame per launch (the latency path): integer descent over a
//                        4-byte-per-slot traversal image (child offset or leaf sigma in one word),
//                        restart from the deepest ancestor shared with the previous step (per-lane
//                        ancestor stack in LDS), 8x8-pixel wave tiles in an XCD-interleaved strip
//                        order, register-resident thresholds/hit lists with static indexing only,
//                        table-driven RNG jump.
//   render_persist<SPP>  N == 2, up to 128 frames per launch (the throughput path): persistent waves,
//                        ray compaction, one ray queue per XCD; sample_kernel before it (thresholds)
//                        and shade_kernel after it (SH colour + pixel epilogue).
// Since round 4 render_fast and render_persist walk the TWO-LEVEL traversal image when the tree has one
// (TreeDev::widew, rto_abi.cpp build_wide_image): top-grid cells and nodes merged with their eight children in
// ONE array, so that a node visit is one uniform 4-byte load resolving two levels; a ray's hit entries name
// entries of that image, wait in LDS while it marches and are translated to leaf slots and written once, when
// it has ended (flush_hits).  The one-level image (nodew + topgrid) serves the counting instantiation and trees
// whose two-level image would not fit; both give the same pixels.
//
// Why the descent can be done on integers (SURVEY.md section 7 "hard parts"): after the clamp to
// [0, 1-1e-6] every operation of the reference descent (x*=2; floor; x-=floor) is exact in fp32,
// so the child digit at level l is bit (23-l) of floor(pos*2^24) and the leaf-local coordinate is
// frac(pos * 2^(l+1)) exactly.
#include <hip/hip_runtime.h>

#include "rto_kernel_types.h"

#pragma clang fp contract(off)

namespace rto {

// ------------------------------------------------------------------ shared pieces

// A pointer loaded from device memory (the frame table's aux / image / hits) is generic to the compiler, which then issues
// flat_load / flat_store for it.  It is device memory: typed as a global-address-space pointer the accesses become
// global_load / global_store.
#define RTO_GLOBAL __attribute__((address_space(1)))
template <class T>
RTO_DEV RTO_GLOBAL T* as_global(T* p) {
    return (RTO_GLOBAL T*)p;
}

// cuda/common.cuh:16-27
RTO_DEV float norm3(const float* d) { return sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); }
RTO_DEV void normalize3(float* d) {
    const float invnorm = 1.f / norm3(d);
    d[0] *= invnorm;
    d[1] *= invnorm;
    d[2] *= invnorm;
}

// volrend.cu:23-56,138-144: pixel -> (dir, vdir, cen) in tree space
RTO_DEV void ray_setup(int x, int y, const CamDev& cam, const TreeDev& tree, float* dir, float* vdir,
                       float* cen) {
    const float xyz[3] = {(x - 0.5f * cam.width) / cam.fx, -(y - 0.5f * cam.height) / cam.fy, -1.0f};
    const float* m = cam.transform;
    dir[0] = m[0] * xyz[0] + m[3] * xyz[1] + m[6] * xyz[2];
    dir[1] = m[1] * xyz[0] + m[4] * xyz[1] + m[7] * xyz[2];
    dir[2] = m[2] * xyz[0] + m[5] * xyz[1] + m[8] * xyz[2];
    normalize3(dir);
    cen[0] = m[9];
    cen[1] = m[10];
    cen[2] = m[11];
    vdir[0] = dir[0];
    vdir[1] = dir[1];
    vdir[2] = dir[2];
    if (tree.ndc_width > 0) {  // maybe_world2ndc :35-56
        const float t = -(1.f + cen[2]) / dir[2];
        for (int i = 0; i < 3; ++i) cen[i] = cen[i] + t * dir[i];
        dir[0] = -((2 * tree.ndc_focal) / tree.ndc_width) * (dir[0] / dir[2] - cen[0] / cen[2]);
        dir[1] = -((2 * tree.ndc_focal) / tree.ndc_height) * (dir[1] / dir[2] - cen[1] / cen[2]);
        dir[2] = -2 / cen[2];
        cen[0] = -((2 * tree.ndc_focal) / tree.ndc_width) * (cen[0] / cen[2]);
        cen[1] = -((2 * tree.ndc_focal) / tree.ndc_height) * (cen[1] / cen[2]);
        cen[2] = 1 + 2 / cen[2];
        normalize3(dir);
    }
    for (int i = 0; i < 3; ++i) cen[i] = tree.offset[i] + tree.scale[i] * cen[i];
}

// rt_core.cuh:206-222: scale dir, invdir, slab test.  returns false when the ray misses the box.
// SEQ: one axis after the other (scheduling barriers): the three double-precision reciprocals and slab tests interleaved keep
// ~30 VGPRs busy, which the reservoir kernel -- it sets a tile up while its lanes hold rays in flight -- does not have
template <bool SEQ = false>
RTO_DEV bool ray_enter(const TreeDev& tree, const OptDev& opt, float* dir, const float* cen, float tmax_bg,
                       float* invdir, float& delta_scale, float& tmin, float& tmax) {
    dir[0] *= tree.scale[0];
    dir[1] *= tree.scale[1];
    dir[2] *= tree.scale[2];
    delta_scale = 1.f / norm3(dir);
    dir[0] *= delta_scale;
    dir[1] *= delta_scale;
    dir[2] *= delta_scale;
    tmax_bg /= delta_scale;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        invdir[i] = 1.f / (dir[i] + 1e-9);  // double
        if (SEQ) __builtin_amdgcn_sched_barrier(0);
    }
    tmin = 0.0;
    tmax = 1e4;
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // _dda_world :19-36, double sub-expressions
        const float t1 = (opt.render_bbox[i] + 1e-6 - cen[i]) * invdir[i];
        const float t2 = (opt.render_bbox[i + 3] - 1e-6 - cen[i]) * invdir[i];
        tmin = f_max(tmin, f_min(t1, t2));
        tmax = f_min(tmax, f_max(t1, t2));
        if (SEQ) __builtin_amdgcn_sched_barrier(0);
    }
    tmax = f_min(tmax, tmax_bg);
    return !(tmax < 0 || tmin > tmax);
}

// min(max(x, 0), 1 - 1e-6) (n3tree_query.hpp:20-24 clamp) as one v_med3_f32: the same value for every
// finite x (the sign of a zero result may differ, which no later operation can observe: the
// fixed-point conversion, fract * invdir and the sums that follow give the same numbers)
RTO_DEV float clamp_unit(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 1.f - 1e-6f); }
// The batched kernels keep a ray's position SCALED by 2^24 (kPos24): cen and dir are multiplied by 2^24 once, at the ray's
// set-up, and cen24 + t * dir24 is then 2^24 times cen + t * dir bit for bit (a power of two commutes with every rounding),
// so the fixed-point coordinates are a bare float -> integer conversion of the clamped sum -- no multiply per march step --
// and the leaf-local point frac(pos * 2^(level + 1)) is frac(pos24 * 2^(level - 23)).
constexpr float kPos24 = 16777216.f;
RTO_DEV float clamp_unit24(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, (1.f - 1e-6f) * kPos24); }

// _dda_unit rt_core.cuh:38-51
RTO_DEV float dda_unit(const float* p, const float* invdir) {
    float tm = 1e4;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t1 = -p[i] * invdir[i];
        const float t2 = t1 + invdir[i];
        tm = f_min(tm, f_max(t1, t2));
    }
    return tm;
}

// lumisphere.hpp:38-80; double literals, one rounding per assignment
RTO_DEV void sh_basis(int basis_dim, const float* dir, float* out) {
    out[0] = 0.28209479177387814;
    const float x = dir[0], y = dir[1], z = dir[2];
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, yz = y * z, xz = x * z;
    switch (basis_dim) {
        case 25:
            out[16] = 2.5033429417967046 * xy * (xx - yy);
            out[17] = -1.7701307697799304 * yz * (3 * xx - yy);
            out[18] = 0.9461746957575601 * xy * (7 * zz - 1.f);
            out[19] = -0.6690465435572892 * yz * (7 * zz - 3.f);
            out[20] = 0.10578554691520431 * (zz * (35 * zz - 30) + 3);
            out[21] = -0.6690465435572892 * xz * (7 * zz - 3);
            out[22] = 0.47308734787878004 * (xx - yy) * (7 * zz - 1.f);
            out[23] = -1.7701307697799304 * xz * (xx - 3 * yy);
            out[24] = 0.6258357354491761 * (xx * (xx - 3 * yy) - yy * (3 * xx - yy));
            [[fallthrough]];
        case 16:
            out[9] = -0.5900435899266435 * y * (3 * xx - yy);
            out[10] = 2.890611442640554 * xy * z;
            out[11] = -0.4570457994644658 * y * (4 * zz - xx - yy);
            out[12] = 0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy);
            out[13] = -0.4570457994644658 * x * (4 * zz - xx - yy);
            out[14] = 1.445305721320277 * z * (xx - yy);
            out[15] = -0.5900435899266435 * x * (xx - 3 * yy);
            [[fallthrough]];
        case 9:
            out[4] = 1.0925484305920792 * xy;
            out[5] = -1.0925484305920792 * yz;
            out[6] = 0.31539156525252005 * (2.0 * zz - xx - yy);
            out[7] = -1.0925484305920792 * xz;
            out[8] = 0.5462742152960396 * (xx - yy);
            [[fallthrough]];
        case 4:
            out[1] = -0.4886025119029199 * y;
            out[2] = 0.4886025119029199 * z;
            out[3] = -0.4886025119029199 * x;
    }
}

// basis for the ray + the basis_minmax mask (rt_core.cuh:277-284)
RTO_DEV void ray_basis(const TreeDev& tree, const OptDev& opt, const float* vdir_in, float* basis_fn) {
#pragma unroll
    for (int i = 0; i < RTO_BASIS_MAX_DEV; ++i) basis_fn[i] = 0.f;
    float vdir[3] = {vdir_in[0], vdir_in[1], vdir_in[2]};
    if (opt.rot_on) {  // rodrigues (volrend.cu:58-73): only the view direction of the SH lookup turns
        const float* k = opt.rot_k;
        const float cross[3] = {k[1] * vdir[2] - k[2] * vdir[1], k[2] * vdir[0] - k[0] * vdir[2],
                                k[0] * vdir[1] - k[1] * vdir[0]};
        const float dot = k[0] * vdir[0] + k[1] * vdir[1] + k[2] * vdir[2];
#pragma unroll
        for (int i = 0; i < 3; ++i)  // float + float, then + (float * float) * double in double, one rounding
            vdir[i] = (float)((double)(vdir[i] * opt.rot_cos + cross[i] * opt.rot_sin) + (double)(k[i] * dot) * opt.rot_omc);
    }
    if (tree.format == 1 /*SH*/) sh_basis(tree.basis_dim, vdir, basis_fn);
#pragma unroll
    for (int i = 0; i < RTO_BASIS_MAX_DEV; ++i)
        if (i < opt.basis_minmax[0] || i > opt.basis_minmax[1]) basis_fn[i] = 0.f;
}

// ray_basis for a tree KNOWN to hold B SH basis functions per channel (the shading kernel's record layouts): the same values in
// basis_fn[0 .. B-1] -- the only ones shade_leaf_packed<3 B + 1> / shade_leaf_quant<B> read -- without the run-time switch over
// the basis size, the 25-entry clear and the 25 mask tests (a third of the ~220 instructions the basis cost per hit entry)
template <int B>
RTO_DEV void ray_basis_sh(const OptDev& opt, const float* vdir_in, float* basis_fn) {
    float vdir[3] = {vdir_in[0], vdir_in[1], vdir_in[2]};
    if (opt.rot_on) {  // (as in ray_basis)
        const float* k = opt.rot_k;
        const float cross[3] = {k[1] * vdir[2] - k[2] * vdir[1], k[2] * vdir[0] - k[0] * vdir[2],
                                k[0] * vdir[1] - k[1] * vdir[0]};
        const float dot = k[0] * vdir[0] + k[1] * vdir[1] + k[2] * vdir[2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            vdir[i] = (float)((double)(vdir[i] * opt.rot_cos + cross[i] * opt.rot_sin) + (double)(k[i] * dot) * opt.rot_omc);
    }
    float full[RTO_BASIS_MAX_DEV];
    sh_basis(B, vdir, full);  // (B is a constant: the switch folds)
#pragma unroll
    for (int i = 0; i < B; ++i) basis_fn[i] = full[i];
    if (opt.basis_minmax[0] > 0 || opt.basis_minmax[1] < B - 1) {  // (uniform; the default options mask nothing)
#pragma unroll
        for (int i = 0; i < B; ++i)
            if (i < opt.basis_minmax[0] || i > opt.basis_minmax[1]) basis_fn[i] = 0.f;
    }
}

// rt_core.cuh:286-325 for one hit leaf: out[0..2] += cnt * sigmoid(<basis, coeffs>), out[3] += cnt.
// The summation order (DC, then the 16..24 group, 9..15, 4..8, 1..3, each left to right) is part of
// the result.
RTO_DEV void shade_leaf(const TreeDev& tree, const uint16_t* __restrict__ tv, const float* basis_fn, float cnt,
                        float* out) {
    const int basis_dim = tree.basis_dim;
    if (basis_dim >= 0) {
        int off = 0;
        float t3[3], o3[3];
#define MUL_BASIS_I(k) (basis_fn[k] * half_bits_to_float(tv[off + (k)]))
        for (int c = 0; c < 3; ++c) {
            float tmp = basis_fn[0] * half_bits_to_float(tv[off]);
            switch (basis_dim) {
                case 25:
                    tmp += MUL_BASIS_I(16) + MUL_BASIS_I(17) + MUL_BASIS_I(18) + MUL_BASIS_I(19) + MUL_BASIS_I(20) +
                           MUL_BASIS_I(21) + MUL_BASIS_I(22) + MUL_BASIS_I(23) + MUL_BASIS_I(24);
                    [[fallthrough]];
                case 16:
                    tmp += MUL_BASIS_I(9) + MUL_BASIS_I(10) + MUL_BASIS_I(11) + MUL_BASIS_I(12) + MUL_BASIS_I(13) +
                           MUL_BASIS_I(14) + MUL_BASIS_I(15);
                    [[fallthrough]];
                case 9:
                    tmp += MUL_BASIS_I(4) + MUL_BASIS_I(5) + MUL_BASIS_I(6) + MUL_BASIS_I(7) + MUL_BASIS_I(8);
                    [[fallthrough]];
                case 4:
                    tmp += MUL_BASIS_I(1) + MUL_BASIS_I(2) + MUL_BASIS_I(3);
            }
            t3[c] = tmp;
            off += basis_dim;
        }
#undef MUL_BASIS_I
        sigmoid_cnt3(t3, cnt, o3);  // out[c] += cnt / (1.f + det_expf(-tmp)), rt_core.cuh:314-318
        for (int c = 0; c < 3; ++c) out[c] += o3[c];
    } else {
        for (int j = 0; j < 3; ++j) out[j] += half_bits_to_float(tv[j]) * cnt;
    }
    out[3] += cnt;
}

// volrend.cu:174-212 (offscreen): background composite, 8 aux planes, RGBA32F image, alpha = 1
RTO_DEV void write_pixel(const FrameOut& fo, int64_t SIZE, int idx, float bg, float* out) {
    const float nalpha = 1.f - out[3];
    const float remain = bg * nalpha;
    out[0] += remain;
    out[1] += remain;
    out[2] += remain;
    float* a = fo.aux + idx;
    a[0] = out[0];
    a[SIZE] = out[1];
    a[2 * SIZE] = out[2];
    a[3 * SIZE] = out[3];
    a[4 * SIZE] = out[0] * out[0];
    a[5 * SIZE] = out[1] * out[1];
    a[6 * SIZE] = out[2] * out[2];
    a[7 * SIZE] = out[3] * out[3];
    reinterpret_cast<float4*>(fo.image)[idx] = make_float4(out[0], out[1], out[2], 1.0f);
}

// ------------------------------------------------------------------ generic kernel (any N)

// n3tree_query.hpp:13-48
RTO_DEV int64_t query_from_root(const TreeDev& tree, float* xyz, float& cube_sz) {
    const float fN = (float)tree.N;
    xyz[0] = f_max(f_min(xyz[0], 1.f - 1e-6f), 0.f);
    xyz[1] = f_max(f_min(xyz[1], 1.f - 1e-6f), 0.f);
    xyz[2] = f_max(f_min(xyz[2], 1.f - 1e-6f), 0.f);
    int64_t ptr = 0;
    cube_sz = fN;
    while (true) {
        float index = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            xyz[i] *= fN;
            const float idx_dimi = floorf(xyz[i]);
            index = index * fN + idx_dimi;
            xyz[i] -= idx_dimi;
        }
        const int64_t sub_ptr = ptr + (int32_t)index;
        const int64_t skip = tree.child[sub_ptr];
        if (skip == 0) return sub_ptr;
        cube_sz *= fN;
        ptr += skip * tree.N3;
    }
}

template <int SPP>
__global__ void __launch_bounds__(256) render_generic(const TreeDev tree, const CamDev cam, const OptDev opt,
                                                       const Pcg32 rng_base, const FrameOut fo) {
    const int64_t SIZE = (int64_t)cam.width * cam.height;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= SIZE) return;
    const int x = idx % cam.width, y = idx / cam.width;
    float out[4] = {0.f, 0.f, 0.f, 0.f};

    if (tree.N > 0) {  // enable_draw volrend.cu:98
        float dir[3], vdir[3], cen[3], invdir[3];
        ray_setup(x, y, cam, tree, dir, vdir, cen);
        Pcg32 rng = rng_base;
        pcg_advance(rng, (int64_t)(idx * SPP));  // volrend.cu:157
        float delta_scale, tmin, tmax;
        if (ray_enter(tree, opt, dir, cen, 1e9f, invdir, delta_scale, tmin, tmax)) {
            // sample_dst rt_core.cuh:67-193
            float dst[SPP + 1];
            for (int n = 1; n <= SPP; ++n) {
                const float tv = -det_log_one_minus(pcg_next_float(rng));
                if (n == 1) {
                    dst[0] = tv;
                } else if (tv <= dst[0]) {
                    for (int i = n - 1; i > 0; i--) dst[i] = dst[i - 1];
                    dst[0] = tv;
                } else {
                    int i = n - 1;
                    while (dst[i - 1] > tv) {
                        dst[i] = dst[i - 1];
                        i--;
                    }
                    dst[i] = tv;
                }
            }
            dst[SPP] = 3.402823466e+38f;

            int64_t tree_vals[SPP];
            float cnts[SPP];
            for (int i = 0; i < SPP; ++i) cnts[i] = 0.f;
            uint32_t spp = 0, sh_nums = 0;
            float src = 0;
            float t = tmin;
            while (t < tmax) {  // rt_core.cuh:241-270
                float pos[3] = {cen[0] + t * dir[0], cen[1] + t * dir[1], cen[2] + t * dir[2]};
                float cube_sz;
                const int64_t leaf = query_from_root(tree, pos, cube_sz);
                const float t_subcube = dda_unit(pos, invdir) / cube_sz;
                const float delta_t = t_subcube + opt.step_size;
                const float sigma = half_bits_to_float(tree.data[leaf * tree.data_dim + tree.data_dim - 1]);
                if (sigma > opt.sigma_thresh) {
                    const float delta = delta_t * delta_scale * sigma;
                    if (src + delta >= dst[spp]) {
                        float& cnt = cnts[sh_nums];
                        tree_vals[sh_nums] = leaf;
                        ++sh_nums;
                        do {
                            ++cnt;
                            ++spp;
                        } while (src + delta >= dst[spp]);
                        if (spp == SPP) break;
                    }
                    src += delta;
                }
                t += delta_t;
            }
            if (sh_nums != 0) {
                float basis_fn[RTO_BASIS_MAX_DEV];
                ray_basis(tree, opt, vdir, basis_fn);
                for (uint32_t i = 0; i < sh_nums; i++)
                    shade_leaf(tree, tree.data + tree_vals[i] * tree.data_dim, basis_fn, cnts[i], out);
                constexpr float INV_SPP = 1.0f / SPP;
                out[0] *= INV_SPP;
                out[1] *= INV_SPP;
                out[2] *= INV_SPP;
                out[3] *= INV_SPP;
            }
        }
    }
    write_pixel(fo, SIZE, idx, opt.background_brightness, out);
}

// ------------------------------------------------------------------ traversal image

// One word per child slot: internal -> child[] value, leaf -> kLeafTag | sigma fp16 bits.
// Derived data (like the reference's commented-out occupancy LUT, n3tree.cpp:206-225): it only
// re-packs what child[]/data[] already say, so traversal decisions cannot change.
__global__ void build_nodew_kernel(const int32_t* __restrict__ child, const uint16_t* __restrict__ data,
                                   int64_t n_slots, int data_dim, uint32_t* __restrict__ nodew,
                                   int* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    const int32_t c = child[i];
    if (c == 0) {
        nodew[i] = kLeafTag | (uint32_t)data[i * data_dim + data_dim - 1];
    } else {
        if (nodew_is_leaf((uint32_t)c)) atomicExch(bad, 1);  // |offset| >= 2^30: not encodable
        nodew[i] = (uint32_t)c;
    }
}

// Aligned copy of the SH coefficients for the shading kernels (TreeDev::shrec): per slot the 3 B coefficients of
// data[] in the same order, zero-padded to shrec_halves(B).  Derived data: the same fp16 values.
__global__ void build_shrec_kernel(const uint16_t* __restrict__ data, int64_t n_slots, int data_dim, int rec,
                                   const uint32_t* __restrict__ recidx, uint16_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per half of a slot's record
    if (i >= n_slots * rec) return;
    const int64_t slot = i / rec;
    const int k = (int)(i - slot * rec);
    int64_t dst = slot;
    if (recidx) {  // compact records (RTO_TREE_COMPACT_RECORDS): only the slots that own one
        const uint32_t r = recidx[slot];
        if (r == kNoRecord) return;
        dst = r;
    }
    out[dst * rec + k] = k < data_dim - 1 ? data[slot * data_dim + k] : (uint16_t)0;
}

// The reference-layout arrays back from the derived ones (rto_abi.cpp ensure_reference_arrays): a tree that renders through
// the fast / batched kernels keeps only nodew + shrec resident; the generic kernel's child[] / data[] are rebuilt on
// first use.  Leaf slots get their exact fp16 values back (coefficients from shrec, sigma from the leaf word); an
// internal slot's sigma -- which no query ever returns -- becomes 0.
__global__ void rebuild_reference_kernel(const uint16_t* __restrict__ shrec, const uint32_t* __restrict__ nodew,
                                         const uint32_t* __restrict__ recidx, int64_t n_slots, int data_dim, int rec,
                                         uint16_t* __restrict__ data, int32_t* __restrict__ child) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per half of data[]
    if (i >= n_slots * data_dim) return;
    const int64_t slot = i / data_dim;
    const int k = (int)(i - slot * data_dim);
    const uint32_t w = nodew[slot];
    const bool leaf = nodew_is_leaf(w);
    int64_t src = slot;
    bool has = true;
    if (recidx) {  // compact records: a slot without one (internal, or a leaf of zero density) reads as zeros
        const uint32_t r = recidx[slot];
        has = r != kNoRecord;
        src = r;
    }
    // (shrec == nullptr: entry-ordered records -- rebuild_reference_wide_kernel fills the coefficients in afterwards)
    data[i] = k < data_dim - 1 ? (has && shrec ? shrec[src * rec + k] : (uint16_t)0) : (leaf ? (uint16_t)(w & 0xffffu) : (uint16_t)0);
    if (k == 0) child[slot] = leaf ? 0 : (int32_t)w;
}

// Top-of-tree shortcut (TreeDev::topgrid): one thread per cell of the 2^G-per-axis grid walks its
// root path over node levels 0..G-1 and records where it ends: {slot | level << kGridSlotBits, nodew[slot]}.
__global__ void build_topgrid_kernel(const uint32_t* __restrict__ nodew, int G, uint2* __restrict__ grid) {
    const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= (1u << (3 * G))) return;
    const uint32_t mask = (1u << G) - 1u;
    const uint32_t cx = key >> (2 * G), cy = (key >> G) & mask, cz = key & mask;
    uint32_t node = 0, slot = 0, w = 0;
    int lvl = 0;
    for (;;) {
        const int sh = G - 1 - lvl;
        const uint32_t ci = (((cx >> sh) & 1u) << 2) | (((cy >> sh) & 1u) << 1) | ((cz >> sh) & 1u);
        slot = node * 8u + ci;
        w = nodew[slot];
        if (nodew_is_leaf(w) || lvl == G - 1) break;
        node += w;
        ++lvl;
    }
    grid[key] = make_uint2(slot | ((uint32_t)lvl << kGridSlotBits), w);
}

// ------------------------------------------------------------------ fast kernel (N == 2)

constexpr int kTileW = 32, kTileH = 8;  // workgroup tile; each wave owns an 8x8 sub-tile

// workgroup -> tile (XCD-interleaved strips, rto_kernel_types.h TileMap)
RTO_DEV bool block_tile(const TileMap& tm, int b, int& tx, int& ty) {
    const int xcd = b & 7, q = b >> 3;
    const int per_strip = tm.strip_rows * tm.tiles_x;
    const int j = q / per_strip, rem = q - j * per_strip;
    ty = (j * 8 + xcd) * tm.strip_rows + rem / tm.tiles_x;
    tx = rem % tm.tiles_x;
    return ty < tm.tiles_y;
}

// Layout of the traversal -> shading hand-off buffer, per frame ("split", round 3): entry 0 of every pixel in a dense plane
// [H*W], entries 1.. pixel-major [H*W][SPP-1] behind it -- a pixel's thresholds / hit list behind entry 0 are one contiguous
// run (20 B at SPP 6), and the thresholds kernel writes entry 0 of an 8x8 tile as eight 32-byte row segments instead of 64
// dwords at a 24-byte stride (0.66 -> 0.43 ms per 100 frames for marks + lists + thresholds).  The planar [SPP][H*W] layout of
// round 1 and the plain pixel-major one of round 2 (shading 1.96 vs 2.17 ms) are history.
template <int SPP>
RTO_DEV uint32_t hit_index(uint32_t pixel, uint32_t i, uint32_t SIZE) {
    return i == 0u ? pixel : SIZE + pixel * (uint32_t)(SPP - 1) + (i - 1u);
}
// distance between entries i and i + 1 of a pixel for i >= 1
RTO_DEV uint32_t hit_stride(uint32_t) { return 1u; }

// Hit list entry: leaf slot in the low hit_slot_bits(SPP) bits, (count - 1) above, kHitValid on top (rto_kernel_types.h).
template <int SPP>
RTO_DEV uint32_t hit_pack(uint32_t slot, uint32_t cnt) { return kHitValid | slot | ((cnt - 1u) << hit_slot_bits(SPP)); }
template <int SPP>
RTO_DEV uint32_t hit_slot(uint32_t h) { return h & ((1u << hit_slot_bits(SPP)) - 1u); }
template <int SPP>
RTO_DEV uint32_t hit_count(uint32_t h) { return ((h & ~kHitValid) >> hit_slot_bits(SPP)) + 1u; }

// The record's DD - 1 coefficients as packed halves al[k >> 1] (half k at its packed position) -> the leaf's contribution.
template <int DD>
RTO_DEV void shade_leaf_words(const uint32_t* al, const float* basis_fn, float cnt, float* out) {
    constexpr int B = (DD - 1) / 3;
    // basis_fn[j] * (float)coefficient k, the half widened by the multiply itself (mul_half_lo / _hi: one instruction, same float)
    auto bc = [&](int j, int k) -> float { return (k & 1) ? mul_half_hi(al[k >> 1], basis_fn[j]) : mul_half_lo(al[k >> 1], basis_fn[j]); };
    float t3[3], o3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int off = c * B;
        float tmp = bc(0, off);
        if constexpr (B >= 25) {
            tmp += bc(16, off + 16) + bc(17, off + 17) + bc(18, off + 18) + bc(19, off + 19) + bc(20, off + 20) + bc(21, off + 21) +
                   bc(22, off + 22) + bc(23, off + 23) + bc(24, off + 24);
        }
        if constexpr (B >= 16) {
            tmp += bc(9, off + 9) + bc(10, off + 10) + bc(11, off + 11) + bc(12, off + 12) + bc(13, off + 13) + bc(14, off + 14) +
                   bc(15, off + 15);
        }
        if constexpr (B >= 9) {
            tmp += bc(4, off + 4) + bc(5, off + 5) + bc(6, off + 6) + bc(7, off + 7) + bc(8, off + 8);
        }
        if constexpr (B >= 4) {
            tmp += bc(1, off + 1) + bc(2, off + 2) + bc(3, off + 3);
        }
        t3[c] = tmp;
    }
    sigmoid_cnt3(t3, cnt, o3);  // out[c] += cnt / (1.f + det_expf(-tmp)), rt_core.cuh:314-318
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] += o3[c];
    out[3] += cnt;
}

// Loads the `DD` fp16 values of one leaf record with aligned dword loads and shades it.
// DD = data_dim (28 for SH9, 49 for SH16); the record starts at a 2-byte aligned address.
// halves per record of the aligned SH-coefficient copy (TreeDev::shrec): the 3 B coefficients in a power-of-two stride,
// so that a record lies in ONE 128-byte line -- SH9 64 B, SH16 128 B.  (Measured, 100 frames of the bench scene: SH16
// shading 2.53 ms from data[], 2.36 with 96-byte records, 2.24 with 128-byte ones; SH25 records -- 150 B -- gain
// nothing at 160 B and lose at 256 B, so SH25 trees are shaded from data[].)
__host__ __device__ constexpr int shrec_halves(int basis_dim) { return 3 * basis_dim * 2 <= 64 ? 32 : 64; }

template <int DD>
RTO_DEV void shade_leaf_packed(const TreeDev& tree, uint32_t slot, const float* basis_fn, float cnt, float* out) {
    constexpr int B = (DD - 1) / 3;
    constexpr int NAL = (DD + 1) / 2;
    uint32_t al[NAL];
    if (B <= 16 && tree.shrec) {
        // the aligned copy: 16-byte loads, every coefficient already at its packed position, one 128-byte line per
        // record (a 98-byte record at a 2-byte aligned address straddles 1.76 lines on average)
        constexpr int NQ = (3 * B * 2 + 15) / 16;  // 16-byte loads that hold coefficients
        // (compact records: the slot's record index first -- one more dependent 4-byte gather per hit leaf)
        const uint32_t ridx = tree.recidx ? tree.recidx[slot] : slot;
        const uint4* __restrict__ q = reinterpret_cast<const uint4*>(tree.shrec + (uint64_t)ridx * shrec_halves(B));
        uint4 v[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) v[i] = q[i];
#pragma unroll
        for (int i = 0; i < NAL; ++i) {
            const uint4& w = v[i >> 2];
            al[i] = (i & 3) == 0 ? w.x : (i & 3) == 1 ? w.y : (i & 3) == 2 ? w.z : w.w;
        }
    } else {
        constexpr int NDW = (DD + 2) / 2;  // dwords covering DD halves at either alignment
        const uint64_t hoff = (uint64_t)slot * DD;  // in halves
        const uint32_t odd = (uint32_t)hoff & 1u;
        const uint32_t* __restrict__ p = reinterpret_cast<const uint32_t*>(tree.data + (hoff - odd));
        uint32_t dw[NDW];
#pragma unroll
        for (int i = 0; i < NDW; ++i) dw[i] = p[i];
        // bring half k of the record to packed position k: a funnel shift by 0 or 16 bits per dword
        // (v_alignbit_b32) instead of extracting every coefficient at both alignments and selecting
        const uint32_t sh = odd * 16u;
#pragma unroll
        for (int i = 0; i < NAL; ++i) al[i] = __builtin_amdgcn_alignbit(i + 1 < NDW ? dw[i + 1] : 0u, dw[i], sh);
    }
    shade_leaf_words<DD>(al, basis_fn, cnt, out);
}

// hit index of the wide image (= the index of the leaf's entry) -> the leaf's slot in data[] / shrec[] (what a hit entry
// names): a grid cell's leaf through wgslot; an entry of a wide node is child a of its octree node (when that is a leaf) or
// child b of that child
RTO_DEV uint32_t wide_to_slot(const TreeDev& tree, uint32_t u) {
    const uint32_t pad = tree.wide_grid_nodes * 64u;
    if (u < pad) return tree.wgslot[u];  // a leaf cell of the top grid
    const uint32_t v = u - pad, wn = v >> 6, x2 = (v >> 4) & 3u, y2 = (v >> 2) & 3u, z2 = v & 3u;
    const uint32_t a = (x2 >> 1) << 2 | (y2 >> 1) << 1 | (z2 >> 1), b = (x2 & 1u) << 2 | (y2 & 1u) << 1 | (z2 & 1u);
    const uint32_t N = tree.worig[wn];
    const uint32_t w0 = tree.nodew[N * 8u + a];
    return nodew_is_leaf(w0) ? N * 8u + a : (N + w0) * 8u + b;
}

// the aligned coefficient records in the order of the two-level image's entries (TreeDev::rec_by_entry): record e = the 3 B
// coefficients of the leaf that entry e of widew names, zero-padded to `rec` halves; entries that are internal nodes (or
// padding) keep zeros.  One thread per half.  Derived data: the same fp16 values.
__global__ void build_shrec_wide_kernel(const TreeDev tree, const uint16_t* __restrict__ data, int64_t n_entries, int rec,
                                        uint16_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries * rec) return;
    const uint32_t e = (uint32_t)(i / rec);
    const int k = (int)(i - (int64_t)e * rec);
    uint16_t v = 0;
    if (nodew_is_leaf(tree.widew[e]) && k < tree.data_dim - 1) {
        const uint32_t pad = tree.wide_grid_nodes * 64u, cells = 1u << (3 * tree.top_levels);
        if (e >= pad || e < cells) v = data[(uint64_t)wide_to_slot(tree, e) * tree.data_dim + k];
    }
    out[i] = v;
}

// ... and back: the coefficients of data[] from entry-ordered records (rebuild_reference_kernel wrote child[], sigma and
// zeros before).  A first-level leaf's 8 entries write the same values to the same place.
__global__ void rebuild_reference_wide_kernel(const TreeDev tree, int64_t n_entries, int rec, uint16_t* __restrict__ data) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries * rec) return;
    const uint32_t e = (uint32_t)(i / rec);
    const int k = (int)(i - (int64_t)e * rec);
    if (k >= tree.data_dim - 1 || !nodew_is_leaf(tree.widew[e])) return;
    const uint32_t pad = tree.wide_grid_nodes * 64u, cells = 1u << (3 * tree.top_levels);
    if (e < pad && e >= cells) return;  // padding behind the grid cells
    data[(uint64_t)wide_to_slot(tree, e) * tree.data_dim + k] = tree.shrec[(uint64_t)e * rec + k];
}

// entry of the two-level image that holds the point (ix, iy, iz) (24-bit fixed point): the walk of render_fast, from the grid
RTO_DEV uint32_t wide_entry_of(const TreeDev& tree, uint32_t ix, uint32_t iy, uint32_t iz) {
    const int G = tree.top_levels;
    uint32_t node = 0u, slot;
    int pr = -1;
    for (;;) {
        const uint32_t b = node ? 2u : (uint32_t)G, msk = (1u << b) - 1u;
        const uint32_t off = node ? (uint32_t)(22 - G - 2 * pr) : 24u - (uint32_t)G;
        slot = (((node << b | ((ix >> off) & msk)) << b | ((iy >> off) & msk)) << b) | ((iz >> off) & msk);
        const uint32_t w = tree.widew[slot];
        if (nodew_is_leaf(w)) return slot;
        node = w;
        ++pr;
    }
}

// STATS: also count the units of SURVEY 8(d)'s algorithmic-byte formula (march steps, descent
// levels a root-restart walk would visit, distinct hit leaves, ...) into fo.stats.  Separate
// instantiation; the timed kernel carries none of it.
// Waves per SIMD render_fast is built for.  A lone frame does not care (it waits for its longest rays on a nearly empty chip);
// callers with several frames in flight do: 5 waves (96 VGPRs; the spills are in the shading tail) lift the pipelined reference
// loop from 5.65 k to 6.07 k frames/s with the sequential one unchanged, 6 waves give 6.1 k and cost the lone frame 1 %
// (profiles/r4_w_ab_fast_wps.txt).  The large-SPP instantiations keep 4 (their threshold / hit arrays live in registers).
#ifndef RTO_FAST_WPS
#define RTO_FAST_WPS 5
#endif
// STACK == 1 (two-level image, at most two pairs of levels below the grid; the launcher decides): the restart of render_persist's
// register-stack form -- the node a step starts from is chosen by where the ray is and which coordinate bits changed
// (rto_march_leaf.inc), positions are kept scaled by 2^24 (kPos24), the step's power-of-two factors come from the level bits of
// the leaf word -- seven dependent instructions fewer on the chain a lone frame's longest rays wait for
template <int SPP, bool STATS, bool WIDE, int STACK = 0>
__global__ void __launch_bounds__(256, SPP <= 8 ? RTO_FAST_WPS : 4) render_fast(const TreeDev tree, const CamDev cam, const OptDev opt,
                                                    const Pcg32 rng_base, const PcgJumpEntry* __restrict__ jump,
                                                    const TileMap tm, const FrameOut fo) {
    extern __shared__ uint32_t s_stack[];  // [max_depth][256] ancestor node indices, level-major

    int tx, ty;
    if (!block_tile(tm, blockIdx.x, tx, ty)) return;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int x = tx * kTileW + wave * 8 + (lane & 7);
    const int y = ty * kTileH + (lane >> 3);
    if (x >= cam.width || y >= cam.height) return;
    const int64_t SIZE = (int64_t)cam.width * cam.height;
    const int idx = y * cam.width + x;

    float out[4] = {0.f, 0.f, 0.f, 0.f};
    if (!STATS && fo.cull_marks) {  // (wave-uniform: a wave is one 8x8 tile)
        const uint32_t t = (uint32_t)(y >> 3) * ((uint32_t)(cam.width + 7) >> 3) + (uint32_t)(x >> 3);
        if (!(((fo.cull_marks[t >> 5] >> (t & 31u)) | fo.cull_marks[fo.cull_mask_words - 1]) & 1u)) {
            write_pixel(fo, SIZE, idx, opt.background_brightness, out);  // no ray of this tile meets density: background
            return;
        }
    }
    float dir[3], vdir[3], cen[3], invdir[3];
    ray_setup(x, y, cam, tree, dir, vdir, cen);
    float delta_scale, tmin, tmax;
    unsigned long long st_steps = 0, st_levels = 0, st_hits = 0, st_inbox = 0, st_grid = 0, st_words = 0, st_wide = 0;
    if (ray_enter(tree, opt, dir, cen, 1e9f, invdir, delta_scale, tmin, tmax)) {
        if (STATS) st_inbox = 1;
        Pcg32 rng = rng_base;
        pcg_advance_tab(rng, (uint32_t)(idx * SPP), jump);

        // thresholds, ascending; dst[0] is always the next one to cross (consumed ones shift out)
        float dst[SPP + 1];
#pragma unroll
        for (int n = 0; n < SPP; ++n) {
            float tv = -det_log_one_minus(pcg_next_float(rng));
#pragma unroll
            for (int i = 0; i < n; ++i) {  // static-index insertion: same sorted array
                const float lo = __builtin_fminf(dst[i], tv), hi = __builtin_fmaxf(dst[i], tv);  // (see sample_kernel)
                dst[i] = lo;
                tv = hi;
            }
            dst[n] = tv;
        }
        dst[SPP] = 3.402823466e+38f;

        uint32_t hits[SPP];
#pragma unroll
        for (int i = 0; i < SPP; ++i) hits[i] = 0;
        uint32_t spp = 0, sh_nums = 0;
        float src = 0;
        float t = tmin;

        uint32_t pix = 0, piy = 0, piz = 0;
        int prev_lvl = 0;
        uint32_t* stack = s_stack + tid;
        const int G = tree.top_levels;  // 0: no top grid
        if (WIDE && G == 0) stack[0] = 0u;
        uint32_t stk0 = 0u, stk1 = 0u;
        const bool regstack = WIDE && (tree.max_depth - G + 1) / 2 <= 2;  // (uniform) pairs of levels below the grid
        const float exit_add[3] = {invdir[0] > 0.f ? invdir[0] : 0.f, invdir[1] > 0.f ? invdir[1] : 0.f, invdir[2] > 0.f ? invdir[2] : 0.f};
        static_assert(STACK == 0 || (WIDE && !STATS), "the register-stack restart is for the two-level image");
        // STACK == 1: the node / bit offset / bits per axis the NEXT step starts from (render_persist's rs.node, rs.woff, rs.wb)
        uint32_t cnode = 0u, coff = 24u - (uint32_t)G, cb = (uint32_t)G;
        const uint32_t tgrid = 1u << (24 - G);
        if constexpr (STACK == 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {  // (kPos24)
                cen[i] *= kPos24;
                dir[i] *= kPos24;
            }
        }

        while (t < tmax) {
            // (round 5: the arithmetic forms of the batched kernel's march step -- one v_med3 per clamp, v_fract, the exit
            //  time as t1 + (invdir > 0 ? invdir : 0), no 1e4 start of the minimum: see rto_march_leaf.inc for why each is the
            //  same number -- a lone frame waits for the DEPENDENT chain of its longest ray, a third of which is this arithmetic:
            //  0.307 -> 0.295 ms per lone 800x800 SPP-6 frame, profiles/r5_w_ab_fast.txt)
            float pos[3];
            uint32_t ix, iy, iz;
            if constexpr (STACK == 1) {
#pragma unroll
                for (int i = 0; i < 3; ++i) pos[i] = clamp_unit24(cen[i] + t * dir[i]);
                ix = (uint32_t)pos[0];
                iy = (uint32_t)pos[1];
                iz = (uint32_t)pos[2];
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) pos[i] = clamp_unit(cen[i] + t * dir[i]);
                ix = (uint32_t)(pos[0] * 16777216.f);
                iy = (uint32_t)(pos[1] * 16777216.f);
                iz = (uint32_t)(pos[2] * 16777216.f);
            }
            // levels whose child digit is unchanged since the previous step
            const uint32_t diff = (ix ^ pix) | (iy ^ piy) | (iz ^ piz);
            int lvl = 0;
            if constexpr (STACK != 1) {
                lvl = __clz((int)diff) - 8;
                lvl = lvl < prev_lvl ? lvl : prev_lvl;
            }
            uint32_t node, w, slot;
            bool have_w = false;
            if constexpr (STACK == 1) {
                // (see rto_march_leaf.inc: the same node while the bits above its index bits are unchanged; back to the grid when a
                //  bit at or above 24 - G differs; else, from the second pair, the first pair's node)
                const bool stay = (diff >> (coff + cb)) == 0u, to_grid = diff >= tgrid;
                node = to_grid ? 0u : (stay ? cnode : stk0);
                uint32_t off = to_grid ? 24u - (uint32_t)G : (stay ? coff : 22u - (uint32_t)G);
                uint32_t b = to_grid ? (uint32_t)G : cb;
                for (;;) {
                    slot = (node << b) | __builtin_amdgcn_ubfe(ix, off, b);
                    slot = (slot << b) | __builtin_amdgcn_ubfe(iy, off, b);
                    slot = (slot << b) | __builtin_amdgcn_ubfe(iz, off, b);
                    w = *(const uint32_t*)((const char*)tree.widew + (uint32_t)(slot << 2));  // (< 2^29 entries: a 32-bit byte offset)
                    if (nodew_is_leaf(w)) break;
                    stk0 = off == 24u - (uint32_t)G ? w : stk0;  // (the first pair's node: the one ancestor a later step may need)
                    node = w;  // the wide node two levels down
                    off -= 2u;
                    b = 2u;
                }
                cnode = node;
                coff = off;
                cb = b;
                (void)have_w;
                (void)stk1;
            } else if constexpr (WIDE) {
                // the two-level image (rto_abi.cpp build_wide_image; round 4): one load per TWO levels below the grid -- a lone
                // frame waits for the dependent-load chains of its longest rays, and this shortens every one of them
                // (node, off): (0, 24 - G) = the top grid, whose cells are indexed by G bits per axis; else the wide node of the
                // pair (G + 2 pr, G + 2 pr + 1), two bits per axis from bit 22 - G - 2 pr on.  One array holds both.
                // With two pairs of levels below the grid at most (regstack: a tree of depth <= G + 4) the ancestor stack is two
                // registers: no LDS round trip on the dependent chain of a step.
                int pr = -1;
                node = 0u;
                if (lvl >= G) {
                    pr = (lvl - G) >> 1;
                    node = regstack ? (pr ? stk1 : stk0) : stack[pr * 256];
                    if (node == 0u) pr = -1;  // (no grid levels, first step: the stack still holds the 0 it was given)
                }
                for (;;) {
                    const uint32_t b = node ? 2u : (uint32_t)G, msk = (1u << b) - 1u;
                    const uint32_t off = node ? (uint32_t)(22 - G - 2 * pr) : 24u - (uint32_t)G;
                    slot = (((node << b | ((ix >> off) & msk)) << b | ((iy >> off) & msk)) << b) | ((iz >> off) & msk);
                    w = *(const uint32_t*)((const char*)tree.widew + (uint32_t)(slot << 2));  // (< 2^29 entries: a 32-bit byte offset)
                    if (nodew_is_leaf(w)) break;
                    node = w;  // the wide node two levels down
                    ++pr;
                    if (regstack) {
                        stk0 = pr == 0 ? node : stk0;
                        stk1 = pr == 0 ? stk1 : node;
                    } else {
                        stack[pr * 256] = node;
                    }
                }
                (void)have_w;
                lvl = (int)((w >> kWideLevelShift) & 31u);  // a leaf word of the wide image carries its level
            } else {
            if (lvl < G) {
                // restart above the shortcut levels: ONE 8-byte lookup replaces the walk over node levels
                // 0..G-1 (a chain of dependent loads -- what a lone frame's long rays wait for) and
                // already carries the word of the slot where that walk ends
                const uint32_t gs = 24u - (uint32_t)G;
                const uint32_t key = (((ix >> gs) << G | (iy >> gs)) << G) | (iz >> gs);
                const uint2 e = tree.topgrid[key];
                slot = e.x & kGridSlotMask;
                lvl = (int)(e.x >> kGridSlotBits);
                node = slot >> 3;
                w = e.y;
                have_w = true;
                if (STATS) ++st_grid;
            } else {
                node = lvl ? stack[lvl * 256] : 0u;
            }
            int st_pair = -1;  // STATS: the pair of levels whose wide node the two-level image would have loaded last
            for (;;) {
                if (!have_w) {
                    const int sh = 23 - lvl;
                    const uint32_t ci = (((ix >> sh) & 1u) << 2) | (((iy >> sh) & 1u) << 1) | ((iz >> sh) & 1u);
                    slot = node * 8u + ci;
                    w = tree.nodew[slot];
                    if (STATS) {
                        ++st_words;
                        // render_persist on the two-level image loads ONE entry per pair of levels (G + 2p, G + 2p + 1)
                        const int pr = (lvl - G) >> 1;
                        if (pr != st_pair) ++st_wide;
                        st_pair = pr;
                    }
                }
                have_w = false;
                if (nodew_is_leaf(w)) break;
                node += w;  // two's complement add of the relative offset
                ++lvl;
                stack[lvl * 256] = node;
            }
            }
            pix = ix;
            piy = iy;
            piz = iz;
            prev_lvl = lvl;
            if (STATS) {
                ++st_steps;
                st_levels += (unsigned)(lvl + 1);
            }

            // cube_sz = 2^(lvl+1) and its reciprocal straight from exponent bits; x / 2^k == x * 2^-k
            // bit for bit (a pure exponent shift, or the same single rounding into the denormals)
            float cube_sz, inv_cube;
            if constexpr (STACK == 1) {  // (positions scaled by 2^24: 2^(level + 1 - 24); the level at the word's exponent bits)
                const uint32_t lvl_bits = w & kWideLevelMask;
                cube_sz = __uint_as_float(lvl_bits + ((uint32_t)(128 - 24) << 23));
                inv_cube = __uint_as_float(((uint32_t)126 << 23) - lvl_bits);
            } else {
                cube_sz = __uint_as_float((uint32_t)(128 + lvl) << 23);
                inv_cube = __uint_as_float((uint32_t)(126 - lvl) << 23);
            }
            float ex[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) ex[i] = -__builtin_amdgcn_fractf(pos[i] * cube_sz) * invdir[i] + exit_add[i];
            const float t_subcube = __builtin_fminf(__builtin_fminf(ex[0], ex[1]), ex[2]) * inv_cube;
            const float delta_t = t_subcube + opt.step_size;
            const float sigma = half_bits_to_float((uint16_t)(w & 0xffffu));
            if (sigma > opt.sigma_thresh) {
                const float delta = delta_t * delta_scale * sigma;
                if (src + delta >= dst[0]) {
                    uint32_t cnt = 0;
                    do {
                        ++cnt;
                        ++spp;
#pragma unroll
                        for (int i = 0; i < SPP; ++i) dst[i] = dst[i + 1];
                    } while (src + delta >= dst[0]);
                    // (the counting instantiation walks the one-level image; a tree whose records follow the two-level image's
                    //  entries needs the leaf's entry there: found by that image's walk -- this kernel is never timed)
                    const uint32_t h = hit_pack<SPP>(!WIDE && tree.rec_by_entry ? wide_entry_of(tree, ix, iy, iz) : slot, cnt);
#pragma unroll
                    for (int i = 0; i < SPP; ++i) hits[i] = (i == (int)sh_nums) ? h : hits[i];
                    ++sh_nums;
                    if (spp == SPP) break;
                }
                src += delta;
            }
            t += delta_t;
        }

        if (STATS) st_hits = sh_nums;
        if (sh_nums != 0) {
            float basis_fn[RTO_BASIS_MAX_DEV];
            ray_basis(tree, opt, vdir, basis_fn);
#pragma unroll
            for (int i = 0; i < SPP; ++i) {
                if (i < (int)sh_nums) {
                    uint32_t slot = hit_slot<SPP>(hits[i]);
                    if constexpr (WIDE)
                        if (!tree.rec_by_entry) slot = wide_to_slot(tree, slot);  // hit index of the wide image -> the leaf's slot
                    const float cnt = (float)hit_count<SPP>(hits[i]);
                    if (tree.format == 1 && tree.data_dim == 28)
                        shade_leaf_packed<28>(tree, slot, basis_fn, cnt, out);
                    else if (tree.format == 1 && tree.data_dim == 49)
                        shade_leaf_packed<49>(tree, slot, basis_fn, cnt, out);
                    else if (tree.format == 1 && tree.data_dim == 76)
                        shade_leaf_packed<76>(tree, slot, basis_fn, cnt, out);
                    else
                        shade_leaf(tree, tree.data + (uint64_t)slot * tree.data_dim, basis_fn, cnt, out);
                }
            }
            constexpr float INV_SPP = 1.0f / SPP;
            out[0] *= INV_SPP;
            out[1] *= INV_SPP;
            out[2] *= INV_SPP;
            out[3] *= INV_SPP;
        }
    }
    write_pixel(fo, SIZE, idx, opt.background_brightness, out);
    if (STATS) {  // order as orc_stats: rays, rays_in_box, steps, levels, hit_leaves, hit_rays
        atomicAdd(fo.stats + 0, 1ULL);
        atomicAdd(fo.stats + 1, st_inbox);
        atomicAdd(fo.stats + 2, st_steps);
        atomicAdd(fo.stats + 3, st_levels);
        atomicAdd(fo.stats + 4, st_hits);
        atomicAdd(fo.stats + 5, st_hits ? 1ULL : 0ULL);
        // the same ray as the batched path sees it: marched only if its 8x8 tile is marked (mark_tiles_kernel); one
        // top-grid entry or one traversal-image word per node visit is exactly what render_persist loads (same restart rule)
        bool marched = true;
        if (fo.stat_marks) {
            const uint32_t t = (uint32_t)(y >> 3) * ((uint32_t)(cam.width + 7) >> 3) + (uint32_t)(x >> 3);
            marched = ((fo.stat_marks[t >> 5] >> (t & 31u)) | fo.stat_marks[fo.stat_mask_words - 1]) & 1u;
        }
        if (marched) {
            atomicAdd(fo.stats + 6, 1ULL);
            atomicAdd(fo.stats + 7, st_steps);
            atomicAdd(fo.stats + 8, st_grid);
            atomicAdd(fo.stats + 9, st_words);
            atomicAdd(fo.stats + 10, st_hits);
            atomicAdd(fo.stats + 11, st_inbox);
            atomicAdd(fo.stats + 12, st_wide);
        }
    }
}

// ------------------------------------------------------------------ persistent kernel (N == 2)
//
// render_persist: the throughput form of render_fast.  One launch renders a BATCH of frames
// (independent poses of the same tree): a frame is only ~10 k waves of very uneven length, so a
// one-frame launch spends most of its time waiting for its longest rays on a nearly empty chip
// (profiles/r1_a_*: mean residency 1.7 k of 8 k wave slots).  Here a fixed grid of persistent
// waves pulls rays from queues that span every frame of the batch:
//   * ray compaction: a lane whose ray ended (all SPP thresholds crossed, left the box, missed)
//     idles only until the wave has REFILL such lanes; then a ballot / mbcnt prefix sum hands each
//     idle lane the next ray of the wave's reservoir (one atomicAdd per 64-256 rays);
//   * rays are queued in 8x8-pixel tile order, so a wave's 64 rays stay spatially coherent; one
//     queue per XCD over interleaved bands of tile rows (rounds 3-5: an angular image wedge each), tile-major across the frames (FrameBatch::qstart);
//   * the end-of-queue drain happens once per batch instead of once per frame.
// Per-ray arithmetic is exactly render_fast's; results are bit-identical.

// the per-ray state that survives between march steps (thresholds live in LDS, hits go straight
// to the hand-off buffer)
struct RayState {
    float cen[3], dir[3], invdir[3];
    float pos[3];  // 2^24 * clamp(cen + t * dir, 0, 1 - 1e-6): the point the next march step starts from (scaled: kPos24)
    float delta_scale, t, tmax, src, cur;      // cur = next threshold to cross (dst[spp])
    uint32_t spp;
    uint32_t pix, piy, piz;
    int prev_lvl;   // level of the node about to be visited
    uint32_t hoff, hnext;  // index of this pixel's next free hit entry in the hand-off buffer, and of the one behind it
                           // (staged hit lists, the default: of its entries 0 and 1, fixed for the ray's life)
    uint32_t nh;           // staged hit lists: hit entries of this ray parked in LDS, not yet written out
    uint32_t node;  // node about to be visited; kGridNext = the top grid is visited next (two-level image: 0 = the grid)
    uint32_t woff;  // two-level image: lowest of the coordinate bits that index the node about to be visited
    uint32_t wb;    // two-level image: coordinate bits per axis that index it -- G at the grid, 2 at a wide node (round 5: kept with
                    // the ray instead of re-derived from `node` by a compare + select in every iteration)
    float cxy __attribute__((ext_vector_type(2)));  // cen[0], cen[1] as a register pair for the packed march arithmetic
    // _dda_unit's max(t1, t1 + invdir) per axis is t1 + (invdir > 0 ? invdir : 0): the sign of invdir is the ray's, not the
    // step's (exit_add below)
    float exit_add[3];
};
constexpr uint32_t kGridNext = 0xffffffffu;
constexpr int kCamFloats = 14;  // fx, fy, transform[12]: what a ray set-up reads of a FrameDesc

// ------------------------------------------------------------------ empty-space culling + ray queues (round 3)
// A ray that never meets a leaf of positive density composites nothing: its pixel is the background, its hit list empty
// (rt_core.cuh:252-262 only ever accumulates in leaves with sigma > sigma_thresh).  73 % of the bench scene's rays are such
// rays and a third of all march steps are theirs.  mark_tiles_kernel decides it per 8x8-pixel tile, conservatively and
// without marching: every culling cell of the tree (TreeDev::occ_cells: world-space bounding spheres of the cubes that hold
// the leaves of positive density, radius padded far above the float error of a sample point) is projected into every
// frame of the batch; the tiles its projection can touch are marked.  An unmarked tile holds no ray that passes within a
// sphere, hence no ray that visits a dense leaf: its rays are never queued, its pixels get no threshold draws.  The bound:
// with the cell centre at camera coordinates (a, b, -d), d > 2 r, every point of the sphere lands within
// f r / (d - r) (1 + |a| / d) pixels of the centre's pixel along x (same with b along y) -- from
// |a'/d' - a/d| <= (r d + |a| r) / (d (d - r)).  A sphere nearer than that marks the whole frame; one behind the
// camera nothing.  (Pixels are bit-identical with and without: tests/test_culling.py, and every parity test against the
// oracle, which marches every ray, runs with it.)
// One workgroup = kMarkCells cells of one frame, marked into a private copy of the frame's mask in LDS (LDS_MASK; a frame
// of more than kMarkLdsWords * 32 tiles marks straight into memory) that is OR-ed into the frame's mask once at the end:
// tens of thousands of cells land on a few hundred mask words, and device-scope atomics on one address serialise
// (the first version, one global atomic per cell and tile, took 2.5 ms per 100 frames; this one 0.1).
constexpr int kMarkCells = 2048, kMarkLdsWords = 8192;
template <class MarkFn>
RTO_DEV void mark_cell(const float4 cell, const FrameDesc& fd, const FrameBatch& fb, MarkFn mark);
template <bool LDS_MASK>
__global__ void __launch_bounds__(256) mark_tiles_kernel(const TreeDev tree, const FrameBatch fb, uint32_t* __restrict__ mask) {
    extern __shared__ uint32_t s_mask[];
    const FrameDesc& fd = fb.f[blockIdx.y];
    uint32_t* const gm = mask + (size_t)blockIdx.y * fb.mask_words;
    if (LDS_MASK) {
        for (int i = threadIdx.x; i < fb.mask_words; i += 256) s_mask[i] = 0u;
        __syncthreads();
    }
    // (round 6: a thread takes a RUN of kMarkCells / 256 consecutive cells, not every 256th: the cells come in Morton order, so the
    //  64 lanes of a wave then mark 64 different neighbourhoods instead of one -- their ds_or_b32 land on different words instead
    //  of serialising on a few: 0.104 -> 0.07 ms per 100 C2 frames, profiles/r6_zz_ab_mark_cells.txt)
    static_assert(kMarkCells % 256 == 0, "a run per thread");
    for (int i = 0; i < kMarkCells / 256; ++i) {
        const int c = (int)blockIdx.x * kMarkCells + (int)threadIdx.x * (kMarkCells / 256) + i;
        if (c >= tree.n_occ_cells) break;
        if (LDS_MASK)
            mark_cell(tree.occ_cells[c], fd, fb, [&](uint32_t w, uint32_t bits) { atomicOr(&s_mask[w], bits); });  // ds_or_b32
        else
            mark_cell(tree.occ_cells[c], fd, fb, [&](uint32_t w, uint32_t bits) { atomicOr(gm + w, bits); });
    }
    if (LDS_MASK) {
        __syncthreads();
        for (int i = threadIdx.x; i < fb.mask_words; i += 256)
            if (s_mask[i]) atomicOr(gm + i, s_mask[i]);
    }
}

template <class MarkFn>
RTO_DEV void mark_cell(const float4 cell, const FrameDesc& fd, const FrameBatch& fb, MarkFn mark) {
    {
    const float* m = fd.transform;  // columns 0..2: camera axes, column 3: centre (common.cuh:29-44)
    const float p[3] = {cell.x - m[9], cell.y - m[10], cell.z - m[11]};
    // camera coordinates (a, b, cc) of the cell centre: M (a, b, cc)^T = p (Cramer; M need not be orthonormal)
    const float c12[3] = {m[4] * m[8] - m[5] * m[7], m[5] * m[6] - m[3] * m[8], m[3] * m[7] - m[4] * m[6]};
    const float c20[3] = {m[7] * m[2] - m[8] * m[1], m[8] * m[0] - m[6] * m[2], m[6] * m[1] - m[7] * m[0]};
    const float c01[3] = {m[1] * m[5] - m[2] * m[4], m[2] * m[3] - m[0] * m[5], m[0] * m[4] - m[1] * m[3]};
    const float det = m[0] * c12[0] + m[1] * c12[1] + m[2] * c12[2];
    const float inv = 1.f / det;
    const float a = (p[0] * c12[0] + p[1] * c12[1] + p[2] * c12[2]) * inv;
    const float b = (p[0] * c20[0] + p[1] * c20[1] + p[2] * c20[2]) * inv;
    const float d = -(p[0] * c01[0] + p[1] * c01[1] + p[2] * c01[2]) * inv;  // depth along the viewing direction (0, 0, -1)
    // |M^-1| <= its Frobenius norm: the sphere's radius in camera coordinates
    const float frob = sqrtf(c12[0] * c12[0] + c12[1] * c12[1] + c12[2] * c12[2] + c20[0] * c20[0] + c20[1] * c20[1] + c20[2] * c20[2] +
                             c01[0] * c01[0] + c01[1] * c01[1] + c01[2] * c01[2]) * fabsf(inv);
    const float r = cell.w * frob * 1.0001f;
    if (d <= -r) return;    // wholly behind the camera (a NaN pose falls through to "keep everything")
    if (!(d > 2.f * r)) {   // too close for the bound: keep every tile of this frame
        mark((uint32_t)fb.mask_words - 1u, 1u);
        return;
    }
    const float xs = a / d, ys = b / d;
    const float k = r / (d - r);
    const float fx = fabsf(fd.fx), fy = fabsf(fd.fy);
    const float pxc = 0.5f * fb.width + fd.fx * xs, pyc = 0.5f * fb.height - fd.fy * ys;  // volrend.cu:139-141 inverted
    const float rx = fx * k * (1.f + fabsf(xs)) + 1.5f, ry = fy * k * (1.f + fabsf(ys)) + 1.5f;
    const int tiles_x = (fb.width + 7) >> 3, tiles_y = (fb.height + 7) >> 3;
    const float x0f = floorf((pxc - rx) * 0.125f), x1f = floorf((pxc + rx) * 0.125f);
    const float y0f = floorf((pyc - ry) * 0.125f), y1f = floorf((pyc + ry) * 0.125f);
    if (!(x1f >= 0.f && y1f >= 0.f && x0f < (float)tiles_x && y0f < (float)tiles_y)) {
        if (!(x0f == x0f && y0f == y0f)) mark((uint32_t)fb.mask_words - 1u, 1u);  // NaN: keep everything
        return;
    }
    const int x0 = x0f < 0.f ? 0 : (int)x0f, x1 = x1f >= (float)tiles_x ? tiles_x - 1 : (int)x1f;
    const int y0 = y0f < 0.f ? 0 : (int)y0f, y1 = y1f >= (float)tiles_y ? tiles_y - 1 : (int)y1f;
    for (int ty = y0; ty <= y1; ++ty)
        for (int tx = x0; tx <= x1; ++tx) {
            const uint32_t t = (uint32_t)(ty * tiles_x + tx);
            mark(t >> 5, 1u << (t & 31u));
        }
    }
}

// the same for ONE frame whose camera arrives as a kernel argument (rto_launch_renderer: no frame table): one workgroup =
// kMarkCells cells, marks OR-ed into `mask` (zeroed on the stream before)
__global__ void __launch_bounds__(256) mark_tiles_one_kernel(const TreeDev tree, const FrameDesc fd, const int width, const int height,
                                                             const int mask_words, uint32_t* __restrict__ mask) {
    extern __shared__ uint32_t s_mask[];
    FrameBatch fb;
    fb.width = width;
    fb.height = height;
    fb.mask_words = mask_words;
    const bool lds = mask_words <= kMarkLdsWords;
    if (lds) {
        for (int i = threadIdx.x; i < mask_words; i += 256) s_mask[i] = 0u;
        __syncthreads();
    }
    for (int i = 0; i < kMarkCells / 256; ++i) {  // (runs of consecutive cells per thread, as in mark_tiles_kernel)
        const int c = (int)blockIdx.x * kMarkCells + (int)threadIdx.x * (kMarkCells / 256) + i;
        if (c >= tree.n_occ_cells) break;
        if (lds)
            mark_cell(tree.occ_cells[c], fd, fb, [&](uint32_t w, uint32_t bits) { atomicOr(&s_mask[w], bits); });
        else
            mark_cell(tree.occ_cells[c], fd, fb, [&](uint32_t w, uint32_t bits) { atomicOr(mask + w, bits); });
    }
    if (lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < mask_words; i += 256)
            if (s_mask[i]) atomicOr(mask + i, s_mask[i]);
    }
}

RTO_DEV bool tile_marked(const FrameBatch& fb, int frame, uint32_t tile_index) {
    const uint32_t* fm = fb.tile_mask + (size_t)frame * fb.mask_words;
    return ((fm[tile_index >> 5] >> (tile_index & 31u)) | fm[fb.mask_words - 1]) & 1u;
}

// Tile slot `pt` of queue k -> is it live, and its list entry {frame << 20 | ty << 10 | tx}.  The queue order: the tiles
// tile_order[qstart[k] .. qstart[k+1]) (ty << 16 | tx; row-major tiles without a table), tile after tile -- the same tile of
// every frame of the batch in turn (neighbouring poses see nearly the same rays through a tile) -- or frame after frame.
RTO_DEV bool queue_slot(const FrameBatch& fb, int k, uint32_t pt, uint32_t& entry) {
    const uint32_t qt0 = (uint32_t)fb.qstart[k], qtiles = (uint32_t)fb.qstart[k + 1] - qt0, n = (uint32_t)fb.n;
    if (pt >= qtiles * n) return false;
    uint32_t tile, frame;
    if (fb.tile_major) {
        const uint32_t t = pt / n;
        frame = pt - t * n;
        tile = qt0 + t;
    } else {
        frame = pt / qtiles;
        tile = qt0 + (pt - frame * qtiles);
    }
    const uint32_t tiles_x = (uint32_t)(fb.width + 7) >> 3;
    uint32_t tx, ty;
    if (fb.tile_order) {
        const uint32_t code = fb.tile_order[tile];
        ty = code >> 16;
        tx = code & 0xffffu;
    } else {
        ty = tile / tiles_x;
        tx = tile - ty * tiles_x;
    }
    entry = frame << 20 | ty << 10 | tx;
    return tile_marked(fb, (int)frame, ty * tiles_x + tx);
}

// queue k of a compaction chunk (the chunks of a queue are consecutive: fb.qchunk)
RTO_DEV int chunk_queue(const FrameBatch& fb, uint32_t chunk) {
    int k = 0;
    while (k + 1 < fb.n_queues && chunk >= (uint32_t)fb.qchunk[k + 1]) ++k;
    return k;
}

__global__ void __launch_bounds__(kQueueChunk) queue_count_kernel(const FrameBatch fb) {
    __shared__ uint32_t s_w[kQueueChunk / 64];
    const int k = chunk_queue(fb, blockIdx.x);
    uint32_t entry;
    const bool live = queue_slot(fb, k, (blockIdx.x - (uint32_t)fb.qchunk[k]) * kQueueChunk + threadIdx.x, entry);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(live);
    if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < kQueueChunk / 64; ++w) t += s_w[w];
        fb.chunk_count[blockIdx.x] = t;
    }
}

// exclusive scan of the chunk counts (a few thousand chunks): one WAVE per queue, lane shuffles, no barrier (round 6: the
// one-workgroup Hillis-Steele scan, queue after queue, was ~320 barrier rounds = 22 us on the launch chain in front of the
// traversal).  The kernel also arms the ray queues (kQueueWords u64: a launch never depends on how the previous one on its
// context ended) -- one fill launch less on that chain.
static_assert(kQueueWords <= 64 * kMaxQueues, "queue words zeroed by the scan's threads");
__global__ void __launch_bounds__(64 * kMaxQueues) queue_scan_kernel(const FrameBatch fb, unsigned long long* __restrict__ queue) {
    if (threadIdx.x < (unsigned)kQueueWords) queue[threadIdx.x] = 0ULL;
    const int k = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    if (k >= fb.n_queues) return;
    const int c1 = fb.qchunk[k + 1];
    uint32_t carry = 0;
    for (int c0 = fb.qchunk[k]; c0 < c1; c0 += 64) {
        const int c = c0 + lane;
        const uint32_t v = c < c1 ? fb.chunk_count[c] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (c < c1) fb.chunk_base[c] = carry + incl - v;
        carry += (uint32_t)__shfl((int)incl, 63, 64);
    }
    if (lane == 0) fb.qcount[k] = carry;
}

__global__ void __launch_bounds__(kQueueChunk) queue_write_kernel(const FrameBatch fb) {
    __shared__ uint32_t s_w[kQueueChunk / 64];
    const int k = chunk_queue(fb, blockIdx.x);
    uint32_t entry = 0;
    const bool live = queue_slot(fb, k, (blockIdx.x - (uint32_t)fb.qchunk[k]) * kQueueChunk + threadIdx.x, entry);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(live);
    if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (!live) return;
    uint32_t before = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += s_w[w];
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    fb.qlist[(size_t)fb.qstart[k] * (uint32_t)fb.n + fb.chunk_base[blockIdx.x] + before + rank] = entry;
}

// sample_dst (rt_core.cuh:67-193) for every pixel of the batch, at full lane utilisation: the RNG
// jump (volrend.cu:157), SPP draws of -log(1-u) and their sort.  The thresholds go to the hand-off
// buffer slots [i][pixel] that the traversal later overwrites with the pixel's hit list.
// Tiles (waves) per workgroup: ONE, as in the shading kernel and for the same reason -- two thirds of the waves find their tile
// unmarked and leave after one load, the others draw for ~400 instructions, and a workgroup frees its slots when its last wave ends
// (4 waves: 0.438 ms per 100 C2 frames for marks + lists + thresholds, 2: 0.436, 1: 0.427; profiles/r6_w_ab_sample_waves.txt)
#ifndef RTO_SAMPLE_WG_WAVES
#define RTO_SAMPLE_WG_WAVES 1
#endif
constexpr int kSampleWaves = RTO_SAMPLE_WG_WAVES;
// tiles per wave of sample_kernel (round 6): with one 8x8 tile per single-wave workgroup a 100-frame launch is a million
// workgroups, two thirds of which leave after their mark load -- the kernel then runs at the rate workgroups are DISPATCHED
// (~2 per clock chip-wide: 0.24 ms before a single draw), not at any arithmetic rate.  A wave takes a strip of consecutive
// tiles, reads their marks in one round trip (lane i: tile i of the strip, ballot) and walks the marked ones.
#ifndef RTO_SAMPLE_TILES
#define RTO_SAMPLE_TILES 8
#endif
constexpr int kSampleTiles = RTO_SAMPLE_TILES;
static_assert(kSampleTiles >= 1 && kSampleTiles <= 64, "a strip's marks come from one ballot");
template <int SPP>
__global__ void __launch_bounds__(64 * kSampleWaves) sample_kernel(const FrameBatch fb, const PcgJumpEntry* __restrict__ jump) {
    // one wave = a strip of 8x8 tiles (row-major tiles): a culled tile costs its wave one bit of a ballot --
    // with one thread per pixel of a scanline nearly every wave held some marched pixel and paid for all the draws
    const uint32_t SIZE = (uint32_t)fb.width * (uint32_t)fb.height;
    const uint32_t tiles_x = (uint32_t)(fb.width + 7) >> 3, tiles_y = (uint32_t)(fb.height + 7) >> 3, n_tiles = tiles_x * tiles_y;
    const uint32_t tile0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * (uint32_t)kSampleWaves + (threadIdx.x >> 6)) * (uint32_t)kSampleTiles));
    if (tile0 >= n_tiles) return;
    const uint32_t lane = threadIdx.x & 63u;
    // a pixel of a culled tile: an empty hit list, no draws (its RNG stream is its own: nobody observes the skipped ones;
    // shade_kernel reads the marks, not a list)
    bool marked = lane < (uint32_t)kSampleTiles && tile0 + lane < n_tiles;
    if (marked && fb.tile_mask) marked = tile_marked(fb, (int)blockIdx.y, tile0 + lane);
    unsigned long long todo = __builtin_amdgcn_ballot_w64(marked);
    if (todo == 0ULL) return;
    const FrameDesc& fd = fb.f[blockIdx.y];
    RTO_GLOBAL uint32_t* const fhits = as_global(fd.hits);
    while (todo) {
        const uint32_t tile = tile0 + (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1ULL;
        const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const uint32_t x = tx * 8u + (lane & 7u), y = ty * 8u + (lane >> 3);
        if (x >= (uint32_t)fb.width || y >= (uint32_t)fb.height) continue;
        const uint32_t idx = y * (uint32_t)fb.width + x;
        Pcg32 rng;
        rng.state = fd.rng_state;
        rng.inc = fd.rng_inc;
        pcg_advance_tab(rng, idx * (uint32_t)SPP, jump);
        float dst[SPP];
#pragma unroll
        for (int n = 0; n < SPP; ++n) {
            float tv = -det_log_one_minus(pcg_next_float(rng));
#pragma unroll
            for (int i = 0; i < n; ++i) {  // static-index insertion: same sorted array
                // (v_min_f32 / v_max_f32: the draws are never NaN, and the only zero a draw can be is -0.0 = -log(1 - 0), so
                //  these return what the reference's `a < b ? a : b` forms do -- in half the instructions)
                const float lo = __builtin_fminf(dst[i], tv), hi = __builtin_fmaxf(dst[i], tv);
                dst[i] = lo;
                tv = hi;
            }
            dst[n] = tv;
        }
        // (-log(1 - 0) = -0.0: the thresholds are only ever compared, so +0.0 serves; its clear top bit is what the
        //  shading kernel ends a hit list on)
#pragma unroll
        for (int i = 0; i < SPP; ++i) fhits[hit_index<SPP>(idx, (uint32_t)i, SIZE)] = __float_as_uint(dst[i]) & ~kHitValid;
    }
}

// Staged hit lists (round 4, VERDICT r3 task 5; -DRTO_HITS_DIRECT restores the store per hit): a ray's hit entries wait in LDS -- in the rows of its threshold column that
// its consumed thresholds left free -- and are written to the hand-off buffer when the ray has ended: entry 0 into the dense
// plane, entries 1.. as one contiguous run (4 * (n - 1) bytes of ONE 32-byte sector for SPP <= 9), back to back, instead of
// one 4-byte store per hit at the moment it happens (71.6 M scattered dwords per 100 frames cost 2.48 GB of line-granular
// HBM writes for 0.29 GB of payload: the L2 had evicted the sector long before the pixel's next entry arrived).
template <int SPP, bool WIDE>
RTO_DEV void flush_hits(RayState& rs, const TreeDev& tree, uint32_t* __restrict__ hits, const float* s_col, uint32_t hstride,
                        bool translate = true) {
    uint32_t e[SPP];
#pragma unroll
    for (int i = 0; i < SPP; ++i) {
        e[i] = 0u;
        if ((uint32_t)i < rs.nh) {
            e[i] = __float_as_uint(s_col[i * 256]);
            if constexpr (WIDE) {
                constexpr uint32_t smask = (1u << hit_slot_bits(SPP)) - 1u;
                // (off the march loop: the ray has ended; !translate: entry-ordered records, TreeDev::rec_by_entry -- the entry IS the record)
                if (translate) e[i] = (e[i] & ~smask) | wide_to_slot(tree, e[i] & smask);
            }
        }
    }
    hits[rs.hoff] = e[0];
    uint32_t* tp = hits + rs.hnext;
#pragma unroll
    for (int i = 1; i < SPP; ++i)
        if ((uint32_t)i < rs.nh) tp[(uint32_t)(i - 1) * hstride] = e[i];
    rs.nh = 0;
}

// REFILL = idle lanes that trigger a retire + refill round
// Flat traversal: one node visit (one load) per lane per loop iteration -- a lane either descends one level or,
// at a leaf, takes its march step and picks the restart node of the next one -- instead of a nested
// "descend until leaf" loop whose trip count is the maximum over the wave (measured: 1.4 loads per
// lane-step on average, but ~4 per wave-step for the slowest lane).
// STACK: where the ancestor stack lives -- 1: two registers (WIDE and at most two pairs of levels below the grid: decided by the
// launcher, so the loop body holds no wave-uniform "which stack?" dispatch: that was 11 scalar instructions per iteration of ~135
// issue slots), 0: the LDS rows
template <int SPP, int REFILL, int WPS, bool WIDE, int STACK>
__global__ void __launch_bounds__(256, WPS) render_persist(const TreeDev tree, const OptDev opt, const FrameBatch fb,
                                                       unsigned long long* __restrict__ queue,
                                                       uint32_t* __restrict__ hits, const uint32_t chunk) {
    // queue[8 + 8k]: next ray of queue k's list (zeroed on the stream before the launch)
    // LDS: [max_depth+1-top_levels][256] ancestor stack | [SPP+1][256] sorted thresholds | frame table
    extern __shared__ uint32_t s_mem[];
    const int tid = threadIdx.x;
    uint32_t* stack = s_mem + tid;  // [level - G][256]
    // levels top_levels.. only.  STACK == 1 (ancestor stack in a register): two rows all the same -- they hold a ray's two
    // hand-off offsets (rs.hoff, rs.hnext: written at the set-up, read at the flush, dead weight in the march loop whose
    // 64-register budget the restart's constants need)
    const int stack_levels = STACK == 1 ? 2 : tree.max_depth + 1 - tree.top_levels;
    float* s_dst = reinterpret_cast<float*>(s_mem + (size_t)stack_levels * 256) + tid;
    // the cameras of the batch: {fx, fy, transform[12]} per frame = the head of a FrameDesc (56 of its 96 bytes: at 100 frames
    // per launch the table then leaves room for 8 workgroups per CU)
    float* s_cams = reinterpret_cast<float*>(s_mem + (size_t)(stack_levels + SPP + 1) * 256);
    __shared__ int s_qstart[kMaxQueues + 1];
    __shared__ uint32_t s_qcount[kMaxQueues];  // live tile slots of each queue (queue_scan_kernel)
    static_assert(offsetof(FrameDesc, transform) == 8 && kCamFloats == 14, "s_cams copies the first 14 floats of a FrameDesc");
    for (int i = tid; i < fb.n * kCamFloats; i += 256) {  // device memory -> LDS
        const int f = i / kCamFloats;
        s_cams[i] = reinterpret_cast<const float*>(fb.f + f)[i - f * kCamFloats];
    }
#pragma unroll
    for (int k = 0; k <= kMaxQueues; ++k)
        if (tid == 64 + k) s_qstart[k] = fb.qstart[k];
#pragma unroll
    for (int k = 0; k < kMaxQueues; ++k)
        if (tid == 128 + k) s_qcount[k] = k < fb.n_queues ? fb.qcount[k] : 0u;
    __syncthreads();

    const int W = fb.width, H = fb.height;
    const uint32_t SIZE = (uint32_t)W * (uint32_t)H;
    const uint32_t hstride = hit_stride(SIZE);  // distance between consecutive entries of one pixel (behind the first)
    // the queue this wave draws from first: the one of the XCD it runs on (HW_REG_XCC_ID bits 3:0)
    const uint32_t n_queues = (uint32_t)fb.n_queues;
    uint32_t cur_q = n_queues > 1 ? ((uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) % n_queues) : 0u;
    uint32_t q_tried = 0;            // queues found empty so far (wave-uniform)
    uint32_t res_off = 0;            // list offset of the queue the reservoir was drawn from (wave-uniform)

    // Loop-invariant scalars pinned in SGPRs: hipcc otherwise re-loads them from the kernarg segment inside the descent loop (an s_load +
    // lgkmcnt(0) round trip per level).
    typedef const __attribute__((address_space(1))) uint32_t* gptr_t;  // keep global_load (not flat_load)
    // WIDE: the two-level image and its top grid (rto_abi.cpp build_wide_image) instead of the one-level ones
    const uint32_t* nodew_p = WIDE ? tree.widew : tree.nodew;
    const uint2* topgrid_p = tree.topgrid;  // (WIDE: unused -- the grid cells are the first entries of the two-level image)
    const uint32_t* __restrict__ qlist = fb.qlist;
    float step_size = opt.step_size, sigma_thresh = opt.sigma_thresh;
    asm volatile("" : "+s"(nodew_p), "+s"(topgrid_p), "+s"(step_size), "+s"(sigma_thresh));
    const gptr_t nodew = (gptr_t)nodew_p;
    typedef unsigned int __attribute__((ext_vector_type(2))) u32x2;
    typedef const __attribute__((address_space(1))) u32x2* gptr2_t;
    const gptr2_t topgrid = (gptr2_t)topgrid_p;
    const int G = tree.top_levels;  // grid bits per axis; the LDS stack holds node levels G.. (entry 0 = level G)
    if (G == 0) stack[0] = 0u;      // no top grid: level 0 is the root
    // indexed by node level (only ever with levels >= G); WIDE: by the PAIR of levels (G + 2p, G + 2p + 1) a wide node spans
    uint32_t* const stack_g = WIDE ? stack : stack - G * 256;

#ifdef RTO_DBG_COUNTERS
    // per-branch occupancy of the march loop (tools/dbg_counters.py): for each branch, how many wave-level executions and
    // how many lanes took part.  0 iteration (any active lane), 1 descend, 2 leaf (march step), 3 sigma > thresh,
    // 4 hit (threshold crossed), 5 restart (ray goes on), 6 ray set-up (refill round), 7 grid lookups
    unsigned dbg_w[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RTO_DBG_AT(i)                                                                                          \
    {                                                                                                          \
        const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true);                                      \
        ++dbg_l[i];                                                                                            \
        if ((tid & 63) == __ffsll((long long)m_) - 1) ++dbg_w[i];                                              \
    }
#else
#define RTO_DBG_AT(i) {}
#endif
    // Two pairs of levels below the grid at most (a tree of depth <= G + 4: the NeRF-synthetic PlenOctrees' 9-10 levels) and the
    // ancestor "stack" is two registers: the restart node then comes from a select, not from an LDS read on the path of
    // every iteration (-2 % in one box, profiles/r4_r_ab_regstack.txt).  Deeper trees keep the LDS rows.  (wave-uniform)
    uint32_t stk0 = 0u, stk1 = 0u;
    static_assert(WIDE || STACK == 0, "the register stack is for the two-level image");
    constexpr bool regstack = STACK == 1, kStackInRegs = regstack;
    uint32_t g_vgpr = (uint32_t)tree.top_levels;  // (rto_march_leaf.inc: the restart's `wb` select)
    asm volatile("" : "+v"(g_vgpr));
    // (the restart's selects, rto_march_leaf.inc: bit offsets of the grid and of the first pair below it, in VGPRs; the
    //  coordinate difference from which a ray is back at the grid, in an SGPR)
    uint32_t woff_grid_v = 24u - (uint32_t)tree.top_levels, woff_pair0_v = 22u - (uint32_t)tree.top_levels;
    uint32_t tgrid = 1u << (24 - tree.top_levels);
    asm volatile("" : "+v"(woff_grid_v), "+v"(woff_pair0_v), "+s"(tgrid));
    constexpr bool kOffsInLds = STACK == 1;
    RayState rs;
    // a lane marches a ray while rs.t < rs.tmax: that comparison IS the lane's state (an ended ray has t >= tmax or
    // tmax = -1), so the wave-level count of marching lanes is the ballot of one v_cmp instead of a loop-carried flag
    rs.t = 0.f;
    rs.tmax = -1.f;
    rs.nh = 0;
    bool drained = false;   // queue exhausted (wave-uniform)
    const uint32_t kChunk = chunk;           // rays per global dequeue (a multiple of the 64-ray tile)
    uint32_t res_next = 0, res_end = 0;      // the wave's private reservoir (wave-uniform)

    // Two nested loops (round 3): the OUTER one refills the wave, the INNER one marches until REFILL lanes are idle again.
    // The inner loop's back edge is one compare + population count + branch; with a single loop that re-decided "refill?"
    // at its top the compiler spent 14 scalar instructions per iteration on that decision -- and scalar instructions come
    // out of the same issue budget as the vector ones (profiles/r3_valu_calibration.json).
    for (;;) {
        {
            for (;;) {
                // (a finished ray needs no retiring: the stale threshold behind its last hit entry ends the list)
                if (drained) break;
                // ---- refill: hand the next queue entries to the idle lanes (ballot + prefix sum)
                const bool idle = !(rs.t < rs.tmax);
                const unsigned long long need = __builtin_amdgcn_ballot_w64(idle);
                const int n_need = __popcll(need);
                if (n_need < REFILL) break;
                // The wave draws rays from a private reservoir [res_next, res_end) and tops it up from
                // the global queue kChunk rays (kChunk/64 tiles) at a time: one device-scope atomic per
                // kChunk rays instead of one per refill (a single counter sustains only ~90 dequeues/us).
                if (res_next == res_end) {
                    for (;;) {  // own queue first, then the others in turn (wave-uniform)
                        const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane(s_qstart[cur_q]);
                        const uint32_t qtotal = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_qcount[cur_q]) * 64u;  // live rays
                        unsigned long long base = 0;
                        if ((tid & 63) == 0) base = atomicAdd(queue + 8 + 8 * cur_q, (unsigned long long)kChunk);
                        const uint32_t base32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
                        if (base32 < qtotal) {  // a counter never exceeds qtotal + kChunk * waves: fits 32 bits
                            res_next = base32;
                            res_end = base32 + kChunk < qtotal ? base32 + kChunk : qtotal;
                            res_off = t0 * (uint32_t)fb.n;
                            break;
                        }
                        if (++q_tried >= n_queues) {
                            drained = true;
                            break;
                        }
                        cur_q = cur_q + 1 == n_queues ? 0u : cur_q + 1;
                    }
                    if (drained) break;
                }
                const uint32_t take = (uint32_t)n_need < res_end - res_next ? (uint32_t)n_need : res_end - res_next;
                const uint32_t first = res_next;
                res_next += take;
                if (idle && rs.nh) {  // the ended ray's hit list leaves in one go
                    if constexpr (kOffsInLds) {
                        rs.hoff = stack[0];
                        rs.hnext = stack[256];
                    }
                    flush_hits<SPP, WIDE>(rs, tree, hits, s_dst, hstride, !tree.rec_by_entry);
                }
                if (idle) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                    const uint32_t r = first + rank;
                    // ray r of the queue: lane (r & 63) of the live tile slot qlist[.. + (r >> 6)] = {frame, tile y, tile x}
                    // (the lists hold marked tiles only: every ray handed out can meet density or at least crosses near it)
                    const uint32_t entry = rank < take ? qlist[res_off + (r >> 6)] : 0u;
                    const int frame = (int)(entry >> 20);
                    const int x = (int)(entry & 1023u) * 8 + (int)(r & 7u);  // (Z-order inside the tile was tried: no fewer L1 accesses)
                    const int y = (int)((entry >> 10) & 1023u) * 8 + (int)((r >> 3) & 7u);
                    if (rank < take && x < W && y < H) {
                        RTO_DBG_AT(6)
                        const float* fd = s_cams + frame * kCamFloats;
                        // (round 5: what the set-up derives from launch constants -- 0.5 W, bbox +- 1e-6 in double, the NDC factors --
                        //  is derived HERE: left alone the compiler hoists those values out of the kernel's loops into ~12 VGPRs and
                        //  spills them; the empty asm statements make the inputs opaque.  This was the kernel's whole private segment.)
                        int Wl = W, Hl = H;
                        OptDev o2 = opt;
                        TreeDev t2 = tree;
                        asm volatile("" : "+s"(Wl), "+s"(Hl));
#pragma unroll
                        for (int i = 0; i < 6; ++i) asm volatile("" : "+s"(o2.render_bbox[i]));
                        asm volatile("" : "+s"(t2.ndc_width), "+s"(t2.ndc_height), "+s"(t2.ndc_focal));
                        CamDev cam;
                        cam.width = Wl;
                        cam.height = Hl;
                        cam.fx = fd[0];
                        cam.fy = fd[1];
#pragma unroll
                        for (int i = 0; i < 12; ++i) cam.transform[i] = fd[2 + i];
                        float vdir[3];
                        ray_setup(x, y, cam, t2, rs.dir, vdir, rs.cen);
                        float tmin;
                        {   // where this pixel's next hit entry goes (hoff) and the one after it (hnext): hit_index
                            const uint32_t fbase = (uint32_t)frame * (uint32_t)SPP * SIZE, pixel = (uint32_t)(y * W + x);
                            rs.hoff = fbase + hit_index<SPP>(pixel, 0u, SIZE);
                            rs.hnext = fbase + hit_index<SPP>(pixel, SPP > 1 ? 1u : 0u, SIZE);
                        }
                        if (ray_enter(t2, o2, rs.dir, rs.cen, 1e9f, rs.invdir, rs.delta_scale, tmin, rs.tmax)) {
                            // sorted thresholds of this pixel (sample_kernel left them in the hand-off
                            // buffer, where the ray's hit list will overwrite them)
                            rs.cur = __uint_as_float(hits[rs.hoff]);
                            const uint32_t* tp = hits + rs.hnext;
                            if constexpr (kOffsInLds) {  // (parked until the ray's flush)
                                stack[0] = rs.hoff;
                                stack[256] = rs.hnext;
                            }
#pragma unroll
                            for (int i = 1; i < SPP; ++i) s_dst[i * 256] = __uint_as_float(tp[(uint32_t)(i - 1) * hstride]);
                            s_dst[SPP * 256] = 3.402823466e+38f;
                            rs.spp = 0;
                            rs.src = 0;
                            rs.t = tmin;
                            float k24 = kPos24;  // (an SGPR operand: as a literal the compiler parks it in a VGPR pair across the kernel)
                            asm volatile("" : "+s"(k24));
#pragma unroll
                            for (int i = 0; i < 3; ++i) {  // (from here on the ray's origin and direction are the scaled ones: kPos24)
                                rs.cen[i] *= k24;
                                rs.dir[i] *= k24;
                            }
                            rs.cxy.x = rs.cen[0];
                            rs.cxy.y = rs.cen[1];
#pragma unroll
                            for (int i = 0; i < 3; ++i) rs.exit_add[i] = rs.invdir[i] > 0.f ? rs.invdir[i] : 0.f;
                            rs.pix = rs.piy = rs.piz = 0;
                            rs.prev_lvl = 0;
                            {  // locate the first position: fixed-point coordinates + first node
#pragma unroll
                                for (int i = 0; i < 3; ++i) rs.pos[i] = clamp_unit24(rs.cen[i] + rs.t * rs.dir[i]);
                                rs.pix = (uint32_t)rs.pos[0];
                                rs.piy = (uint32_t)rs.pos[1];
                                rs.piz = (uint32_t)rs.pos[2];
                                rs.node = WIDE ? 0u : (G > 0 ? kGridNext : 0u);
                                rs.woff = 24u - (uint32_t)G;
                                rs.wb = (uint32_t)G;
                            }
                        } else {
                            rs.tmax = -1.f;  // missed the box (ray_enter wrote a tmax that the stale t might undercut)
                        }
                    }
                }
            }
        }
        bool active = rs.t < rs.tmax;
        if (__builtin_amdgcn_ballot_w64(active) == 0ULL) {
            if (drained) break;
            continue;  // (every ray of the round missed the volume)
        }
        // once the queues are empty there is nothing to refill with: march until the last ray ends
        const int exit_at = drained ? 0 : 64 - REFILL;
        int n_active;
        do {
        {
            // ---- one node visit for every active lane
            if (active) {
                RTO_DBG_AT(0)
                uint32_t slot, w;
                if constexpr (WIDE) {
                    // Round 4: ONE array holds the top grid and the two-level ("wide") nodes below it (rto_abi.cpp
                    // build_wide_image), so a node visit is ONE uniform load: entry = ((node << b | x bits) << b | y bits) << b |
                    // z bits, b bits per axis from bit rs.woff on -- (node, b, woff) = (0, G, 24 - G) at the grid,
                    // (node number, 2, 22 - G - 2 p) at the wide node of the levels (G + 2p, G + 2p + 1).  v_bfe_u32 and
                    // v_lshl_or_b32 take the per-lane widths: no grid / node case split, no second address, no branch pair
                    // around two loads (the one-level walk below spends 19 VALU + 7 SALU where this spends 10 VALU).
                    const uint32_t b = rs.wb;
                    slot = (rs.node << b) | __builtin_amdgcn_ubfe(rs.pix, rs.woff, b);
                    slot = (slot << b) | __builtin_amdgcn_ubfe(rs.piy, rs.woff, b);
                    slot = (slot << b) | __builtin_amdgcn_ubfe(rs.piz, rs.woff, b);
                    if (rs.node == 0u) { RTO_DBG_AT(7) }
#ifdef RTO_STUB_LOADS
                    {   // calibration build (tools/calibrate_valu.sh): the gather replaced by a hash of its address
                        const uint32_t hsh = slot * 0x9E3779B1u;
                        const uint32_t sg = (hsh & 0x600u) ? 0u : 0x4D00u;
                        if (rs.node == 0u) {
                            const uint32_t glv = 2u + (hsh >> 30);
                            w = (glv == 5u && (hsh & 0x100u)) ? ((hsh >> 8) & 0xffffu) | 1u : (kLeafTag | glv << kWideLevelShift | sg);
                        } else {  // two pairs below the grid (levels G .. G + 3), leaves at either level of a pair
                            const uint32_t lv = 22u - rs.woff + ((hsh >> 27) & 1u);
                            w = (rs.woff == 22u - (uint32_t)G && (hsh >> 29) < 5u) ? ((hsh >> 8) & 0xffffu) | 1u : (kLeafTag | lv << kWideLevelShift | sg);
                        }
                    }
#else
                    // (through the L1: non-temporal loads cost 15-50 %.  The byte offset as a 32-bit value -- the image has < 2^29
                    //  entries -- lets the load take its base from SGPRs and one VGPR of offset: no 64-bit address pair, no register
                    //  pinned to zero for its high half)
                    w = *(gptr_t)((const __attribute__((address_space(1))) char*)nodew + (uint32_t)(slot << 2));
#endif
                    if ((int32_t)w >= -(1 << 30)) {  // internal: two levels down (from the grid: into the level-G node)
                        RTO_DBG_AT(1)
                        rs.node = w;
                        if (regstack) {
                            // two pairs at most: the ancestor "stack" is ONE register, the node of the first pair.  (The second
                            // pair's node needs none: a restart inside the second pair happens at a leaf of that very node --
                            // the ray stays in rs.node.)
                            const bool first = rs.woff == 24u - (uint32_t)G;
                            stk0 = first ? w : stk0;
                        } else
                            stack[(((24u - (uint32_t)G) - rs.woff) >> 1) * 256u] = w;  // row p + 1 of the pair it spans (grid: row 0)
                        rs.woff -= 2u;
                        rs.wb = 2u;
                    }
                }
                if constexpr (!WIDE) {
                const bool grid = rs.node == kGridNext;
                const uint32_t gs = 24u - (uint32_t)G;
                const uint32_t key = (((rs.pix >> gs) << G | (rs.piy >> gs)) << G) | (rs.piz >> gs);
                const uint32_t sh = 23u - (uint32_t)rs.prev_lvl;
                slot = (rs.node << 1) | __builtin_amdgcn_ubfe(rs.pix, sh, 1u);  // node * 8 + child digit,
                slot = (slot << 1) | __builtin_amdgcn_ubfe(rs.piy, sh, 1u);       // three v_lshl_or
                slot = (slot << 1) | __builtin_amdgcn_ubfe(rs.piz, sh, 1u);
                // Both addresses exist in registers before either load is issued.  Left to itself the compiler sinks each
                // address computation into its branch, and when a temporary of the second branch lands in the register the
                // first branch's load is still writing, it has to put an s_waitcnt vmcnt(0) between the two loads -- the
                // iteration then pays two memory latencies back to back (measured: 8.16 instead of 7.16 ms per 100
                // frames from a one-instruction difference elsewhere in the kernel that renumbered the registers).
                gptr_t pn = nodew + slot;
                gptr2_t pg = topgrid + key;
                asm volatile("" : "+v"(pn), "+v"(pg));
#ifdef RTO_STUB_LOADS
                // Calibration build only (tools/calibrate_valu.sh): both gathers replaced by a hash of their address -- a
                // procedural stand-in for the tree with the same loop, the same divergence and no memory latency, to measure
                // what the loop body sustains in VALU instructions per clock at 1..8 waves per SIMD.  Never shipped.
                if (grid) {
                    const uint32_t hsh = key * 0x9E3779B1u;
                    const uint32_t glv = 2u + (hsh >> 30);
                    slot = (key << 3) & kGridSlotMask;
                    const uint32_t sg = (hsh & 0x600u) ? 0u : 0x4D00u;
                    rs.prev_lvl = (int)glv;
                    rs.node = slot >> 3;
                    w = (glv == 5u && (hsh & 0x100u)) ? 1u : (kLeafTag | sg);
                } else {
                    const uint32_t hsh = slot * 0x9E3779B1u;
                    const uint32_t sg = (hsh & 0x600u) ? 0u : 0x4D00u;
                    w = (rs.prev_lvl < 9 && (hsh >> 29) < 3u) ? 1u : (kLeafTag | sg);
                }
#else
                if (grid) {  // the iteration's one load: 8 bytes of the top grid ...
                    const u32x2 e = *pg;  // (through the L1 as well: non-temporal costs 15 %)
                    slot = e.x & kGridSlotMask;
                    rs.prev_lvl = (int)(e.x >> kGridSlotBits);
                    rs.node = slot >> 3;
                    w = e.y;
                } else {  // ... or 4 bytes of the traversal image
                    w = *pn;  // (through the L1: a non-temporal load here costs 50 %)
                }
#endif
                if (grid) { RTO_DBG_AT(7) }
                if ((int32_t)w >= -(1 << 30)) {  // internal: one level down
                    RTO_DBG_AT(1)
                    rs.node += w;
                    ++rs.prev_lvl;
                    stack_g[rs.prev_lvl * 256] = rs.node;
                }
                }  // (!WIDE)
                if ((int32_t)w < -(1 << 30)) {  // leaf: the march step (rt_core.cuh:241-270)
#include "rto_march_leaf.inc"
                }
            }
        }
            active = rs.t < rs.tmax;
            {
                const unsigned long long am = __builtin_amdgcn_ballot_w64(active);
                asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n_active) : "s"(am) : "scc");
            }
        } while (n_active > exit_at);
    }
    if (rs.nh) {  // rays that ended after the last refill round
        if constexpr (kOffsInLds) {
            rs.hoff = stack[0];
            rs.hnext = stack[256];
        }
        flush_hits<SPP, WIDE>(rs, tree, hits, s_dst, hstride, !tree.rec_by_entry);
    }
#ifdef RTO_DBG_COUNTERS
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // queue words 1..7 / 9..15 are padding of the queue counters: wave counts, lane counts
        const int wi = i == 0 ? 16 + 1 : i, li = i == 0 ? 16 + 2 : 8 + i;
        if (dbg_w[i]) atomicAdd(queue + wi, (unsigned long long)dbg_w[i]);
        if (dbg_l[i]) atomicAdd(queue + li, (unsigned long long)dbg_l[i]);
    }
#endif
}


// One hit leaf of a quantised tree, shaded straight from the codebooks (TreeDev::qrec / qcolors): the
// coefficients are the very fp16 values N3Tree::load_npz would have expanded (n3tree.cpp:310-339),
// summed in shade_leaf's order, so the pixel is bit-identical to rendering the decoded tree.
template <int B>
RTO_DEV void shade_leaf_quant(const TreeDev& tree, uint32_t slot, const float* basis_fn, float cnt, float* out) {
    const int nr = tree.q_retain;
    const uint16_t* __restrict__ rec = tree.qrec + (uint64_t)slot * (uint32_t)tree.q_rec;
    float v[B][3];
#pragma unroll
    for (int k = 0; k < B; ++k) {
        if (k < nr) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[k][c] = half_bits_to_float(rec[k * 3 + c]);
        } else {
            const uint32_t id = rec[2 * nr + k];  // 3 * nr + (k - nr)
            const uint2 e = tree.qcolors[(uint32_t)(k - nr) * 65536u + id];
            v[k][0] = half_bits_to_float((uint16_t)(e.x & 0xffffu));
            v[k][1] = half_bits_to_float((uint16_t)(e.x >> 16));
            v[k][2] = half_bits_to_float((uint16_t)(e.y & 0xffffu));
        }
    }
    float t3[3], o3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float tmp = basis_fn[0] * v[0][c];
        if constexpr (B >= 25) {
            tmp += basis_fn[16] * v[16][c] + basis_fn[17] * v[17][c] + basis_fn[18] * v[18][c] +
                   basis_fn[19] * v[19][c] + basis_fn[20] * v[20][c] + basis_fn[21] * v[21][c] +
                   basis_fn[22] * v[22][c] + basis_fn[23] * v[23][c] + basis_fn[24] * v[24][c];
        }
        if constexpr (B >= 16) {
            tmp += basis_fn[9] * v[9][c] + basis_fn[10] * v[10][c] + basis_fn[11] * v[11][c] +
                   basis_fn[12] * v[12][c] + basis_fn[13] * v[13][c] + basis_fn[14] * v[14][c] +
                   basis_fn[15] * v[15][c];
        }
        if constexpr (B >= 9) {
            tmp += basis_fn[4] * v[4][c] + basis_fn[5] * v[5][c] + basis_fn[6] * v[6][c] + basis_fn[7] * v[7][c] +
                   basis_fn[8] * v[8][c];
        }
        if constexpr (B >= 4) {
            tmp += basis_fn[1] * v[1][c] + basis_fn[2] * v[2][c] + basis_fn[3] * v[3][c];
        }
        t3[c] = tmp;
    }
    sigmoid_cnt3(t3, cnt, o3);  // out[c] += cnt / (1.f + det_expf(-tmp)), rt_core.cuh:314-318
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] += o3[c];
    out[3] += cnt;
}

// quant_map [nq][ns] + data_retained [nr][ns][3]  ->  slot-major records (TreeDev::qrec)
__global__ void pack_quant_kernel(const uint16_t* __restrict__ qmap, const uint16_t* __restrict__ retained,
                                  int64_t ns, int nr, int nq, int rec, uint16_t* __restrict__ out) {
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= ns) return;
    uint16_t* o = out + slot * rec;
    for (int k = 0; k < nr; ++k)
        for (int c = 0; c < 3; ++c) o[k * 3 + c] = retained[((int64_t)k * ns + slot) * 3 + c];
    for (int j = 0; j < nq; ++j) o[3 * nr + j] = qmap[(int64_t)j * ns + slot];
    if (rec > 3 * nr + nq) o[rec - 1] = 0;
}

hipError_t launch_pack_quant(const uint16_t* qmap, const uint16_t* retained, int64_t ns, int nr, int nq, int rec,
                             uint16_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(pack_quant_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, qmap, retained, ns, nr,
                       nq, rec, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ shading kernel
// Second half of trace_ray (rt_core.cuh:272-331) + the pixel epilogue (volrend.cu:174-212) for the
// batched path.  hits: packed entries per frame, ended by the first word without kHitValid (layout: hit_index).
//
// Only about a quarter of the pixels hit anything and those that do hold 1..SPP leaves, so a
// thread-per-pixel loop leaves most lanes idle while the gathers of a few run.  Instead each wave
// owns 64 * P consecutive pixels, compacts their hit entries into an LDS list (wave prefix sum),
// shades the entries one per lane -- every lane busy, all of an entry's loads independent -- and
// the pixel lanes then add their entries' contributions up in hit order, which keeps the float sums
// those of the reference's loop.  The list is processed in windows of kShadeCap entries so LDS use
// does not depend on SPP.
#ifndef RTO_SHADE_CAP
#define RTO_SHADE_CAP 320
#endif
constexpr int kShadeCap = RTO_SHADE_CAP;

// contribution of one hit leaf: rgb[c] = cnt * sigmoid(<basis, coeffs_c>) (or cnt * rgb for RGBA trees)
// MODE (host-chosen, so that each instantiation carries one leaf layout's registers only):
// 0 any dense tree; 28 / 49 / 76 dense SH9 / SH16 / SH25 records; -B quantised SH<B>, not expanded
template <int MODE>
RTO_DEV void leaf_contrib(const TreeDev& tree, uint32_t slot, const float* basis_fn, float cnt, float* o) {
    o[0] = o[1] = o[2] = o[3] = 0.f;  // 0 + x == x: the helpers' "+=" yields the bare term
    if constexpr (MODE < 0) {
        shade_leaf_quant<-MODE>(tree, slot, basis_fn, cnt, o);
    } else if constexpr (MODE == 0) {
        shade_leaf(tree, tree.data + (uint64_t)slot * tree.data_dim, basis_fn, cnt, o);
    } else {
        shade_leaf_packed<MODE>(tree, slot, basis_fn, cnt, o);
    }
}

#ifndef RTO_SHADE_WPS
#define RTO_SHADE_WPS 4
#endif
// workgroups per CU the kernel is built for: the SH25 layouts (76 coefficients in registers) and the quantised SH16 one need
// more than the 128 VGPRs that 4 leave them (they spilled 10-50 registers to scratch: SH25 shading 9.6 -> 8.7 ms per 100
// frames of 1920x1080 without the spills, at 3)
#ifndef RTO_SHADE_WPS_SH25
#define RTO_SHADE_WPS_SH25 3
#endif
constexpr int shade_wps(int mode) { return (mode == 76 || mode == -25 || mode == -16) ? RTO_SHADE_WPS_SH25 : RTO_SHADE_WPS; }
#ifdef RTO_DBG_COUNTERS
// where a shading wave's lifetime goes (tools/dbg_shade_phases.py): per wave (blockIdx.x * 4 + wave, up to 2^19 of them) the s_memtime
// stamps 0..5 and its number of hit entries; plain stores, reduced on the host (atomics would slow the very kernel they time)
constexpr int kShadeStampWaves = 1 << 19;
__device__ unsigned long long g_shade_phase[kShadeStampWaves * 8];
#define RTO_SHADE_STAMP(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ph[i] = __builtin_amdgcn_s_memtime(); }
#else
#define RTO_SHADE_STAMP(i) {}
#endif
// Waves per workgroup: ONE (round 6).  The waves of this kernel never talk to each other (wave-level barriers only), and their
// lifetimes differ by a factor of eight -- a wave over culled tiles ends after ~7 k clocks, one with 500 hit entries after ~58 k
// (tools/dbg_shade_phases.py, profiles/r6_d_shade_phases.txt) -- but a workgroup's LDS and wave slots are released only when its
// LAST wave ends: with 4 waves per workgroup the SIMDs held 2.9 waves of the 4 they have room for.
#ifndef RTO_SHADE_WG_WAVES
#define RTO_SHADE_WG_WAVES 1
#endif
constexpr int kShadeWaves = RTO_SHADE_WG_WAVES;
#ifndef RTO_SHADE_BAND_ROWS
#define RTO_SHADE_BAND_ROWS 8
#endif
// pixel blocks (of block_px consecutive pixels, scanline order) per band of RTO_SHADE_BAND_ROWS rows
__host__ __device__ constexpr uint32_t shade_blocks_per_band(int W, int block_px) {
    return (uint32_t)((RTO_SHADE_BAND_ROWS > 0 ? RTO_SHADE_BAND_ROWS : 1) * W + block_px - 1) / (uint32_t)block_px;
}
template <int SPP, int P, int MODE>
__global__ void __launch_bounds__(64 * kShadeWaves, shade_wps(MODE)) shade_kernel(const TreeDev tree, const OptDev opt, const FrameBatch fb,
                                                                              const uint32_t* __restrict__ hits0) {
    __shared__ uint32_t s_h[kShadeWaves][kShadeCap];       // packed hit entry
    __shared__ uint16_t s_q[kShadeWaves][kShadeCap];       // its pixel, relative to the wave's first pixel
    __shared__ float s_c[kShadeWaves][3 * kShadeCap];      // its contribution, [channel][entry]
    const int W = fb.width, H = fb.height;
    const int64_t SIZE = (int64_t)W * H;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#ifdef RTO_DBG_COUNTERS
    unsigned long long ph[8];
    for (int i = 0; i < 8; ++i) ph[i] = 0;
    RTO_SHADE_STAMP(0)
#endif
    // Workgroup -> (pixel block, frame), XCD-aware: workgroups go round-robin over the 8 XCDs, so id & 7 picks the XCD, and the L2 is
    // per XCD.  Rounds 2-5 put the same 128-pixel block of ALL frames of the batch on one XCD, frame after frame, hoping that
    // neighbouring poses hit the same leaves there; the counters never agreed (L2 hit 15 %, as with frame-major order): an orbiting
    // camera moves a leaf SIDEWAYS by ten or twenty pixels per frame -- out of its block after a few frames, but not out of its rows.
    // Round 6: BANDS of 8 rows.  Band b of all frames runs on XCD b & 7, frame after frame, a band's blocks side by side: L2 hit
    // 0.15 -> 0.28, FETCH 4.25 -> 3.56 GB per 100 C2 frames, 1.335 -> 1.298 ms (C4 6.75 -> 6.23, C5 0.985 -> 0.936); bands of 4 / 16 /
    // 32 rows: 1.306 / 1.315 / 1.43 (profiles/r6_y_ab_shade_bands*.txt, r6_y_pmc_shade_bands.txt).  RTO_SHADE_BAND_ROWS=0: the old order.
    const uint32_t bid = blockIdx.x, q = bid >> 3;
#if RTO_SHADE_BAND_ROWS > 0
    const uint32_t cpb = shade_blocks_per_band(W, 64 * kShadeWaves * P), per_band = cpb * (uint32_t)fb.n;
    const uint32_t bi = q / per_band, rem = q - bi * per_band;
    const uint32_t frame = rem / cpb;
    const uint32_t pblock = (bi * 8u + (bid & 7u)) * cpb + (rem - frame * cpb);
#else
    const uint32_t frame = q % (uint32_t)fb.n;
    const uint32_t pblock = (q / (uint32_t)fb.n) * 8u + (bid & 7u);
#endif
    const int64_t wave_px0 = ((int64_t)pblock * kShadeWaves + wv) * (64 * P);
    if (wave_px0 >= SIZE) return;  // wave-uniform
    const FrameDesc& fd = fb.f[frame];  // block-uniform index: scalar loads
    // (the frame's hand-off lists: from the launch's base, not from the frame table -- a wave's first loads then depend on its
    //  kernel arguments only; a shading wave spends a fifth of its life before its hit lists have arrived, profiles/r6_d_shade_phases.txt)
    const RTO_GLOBAL uint32_t* const fhits = as_global(hits0) + (size_t)frame * (size_t)SPP * (size_t)SIZE;

    // ---- each lane: the hit lists of its P pixels (pixel p*64 + lane of the wave: coalesced)
    uint32_t h[P][SPP];
    uint32_t n[P];
    uint32_t mine = 0, live_bits = 0;  // bit p: pixel p of this lane lies in a marked tile
    // A pixel of a culled tile has no hit list (nobody wrote one: sample_kernel, render_persist): it is read off the tile
    // marks, a few hundred cached words per frame, instead of 4 * SPP bytes per pixel of stale memory -- two thirds of the
    // pixels of the bench scene.  (x, y) of the wave's first pixel by one wave-uniform division, the lanes' by carries.
    const uint32_t* fmask = fb.tile_mask ? fb.tile_mask + (size_t)frame * fb.mask_words : nullptr;
    const uint32_t keep_all = fmask ? fmask[fb.mask_words - 1] & 1u : 1u;
    const int tiles_x = (W + 7) >> 3;
    const int wy0 = (int)(wave_px0 / W), wx0 = (int)(wave_px0 - (int64_t)wy0 * W);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int64_t idx = wave_px0 + p * 64 + lane;
        n[p] = 0;
        bool live = idx < SIZE;
        if (live && fmask) {  // (the mark word is requested before the frame's keep-all flag is back: no dependent second trip)
            int x = wx0 + p * 64 + lane, y = wy0;
            while (x >= W) {
                x -= W;
                ++y;
            }
            const uint32_t t = (uint32_t)((y >> 3) * tiles_x + (x >> 3));
            live = (((fmask[t >> 5] >> (t & 31u)) | keep_all) & 1u) != 0u;
        }
        live_bits |= live ? 1u << p : 0u;
        if (live) {
            bool open = true;
            // the first entry from its dense plane, the run behind it (fetched eagerly: fetching the run only for a pixel whose
            // first entry is valid saves 20 bytes per empty pixel and costs a dependent round trip per wave, 2.03 vs 1.95 ms)
            uint32_t raw[SPP];
            raw[0] = fhits[idx];
            const RTO_GLOBAL uint32_t* hp = fhits + SIZE + idx * (SPP - 1);
#pragma unroll
            for (int i = 1; i < SPP; ++i) raw[i] = hp[i - 1];
#pragma unroll
            for (int i = 0; i < SPP; ++i) {
                open = open && (raw[i] & kHitValid) != 0u;
                h[p][i] = open ? raw[i] : 0u;
                n[p] += open ? 1u : 0u;
            }
        }
        mine += n[p];
    }
    // (sparse lean outputs: a wave whose pixels all lie in unmarked tiles -- two thirds of the bench scene's waves -- has nothing to
    //  shade and nothing to store)
    if (fb.lean == 2 && __builtin_amdgcn_ballot_w64(live_bits != 0u) == 0ULL) return;
    // ---- wave exclusive prefix sum -> each pixel's range in the compacted list
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    const uint32_t total = __shfl(inc, 63, 64);
    RTO_SHADE_STAMP(1)  // tile marks + hit lists are here (the prefix sum consumed them)
    uint32_t start[P];
    {
        uint32_t sacc = inc - mine;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            start[p] = sacc;
            sacc += n[p];
        }
    }
    float out[P][4];
#pragma unroll
    for (int p = 0; p < P; ++p) out[p][0] = out[p][1] = out[p][2] = out[p][3] = 0.f;

    CamDev cam;
    cam.width = W;
    cam.height = H;
    cam.fx = fd.fx;
    cam.fy = fd.fy;
#pragma unroll
    for (int i = 0; i < 12; ++i) cam.transform[i] = fd.transform[i];

    for (uint32_t w0 = 0; w0 < total; w0 += kShadeCap) {  // wave-uniform
        // ---- pixel lanes publish the entries that fall into this window
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int i = 0; i < SPP; ++i) {
                const uint32_t pos = start[p] + i - w0;  // wraps to a huge value below the window
                if ((uint32_t)i < n[p] && pos < (uint32_t)kShadeCap) {
                    s_h[wv][pos] = h[p][i];
                    s_q[wv][pos] = (uint16_t)(p * 64 + lane);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- one entry per lane
        RTO_SHADE_STAMP(2)  // (last window's) entries published
        const uint32_t cnt_w = min(total - w0, (uint32_t)kShadeCap);
        for (uint32_t j = lane; j < cnt_w; j += 64) {
            const uint32_t he = s_h[wv][j];
            // the entry's pixel = pixel s_q of the wave's run, which starts at (wx0, wy0): carries instead of a 64-bit division
            // per entry (round 5: `idx % W`, `idx / W` on an int64 were ~150 of the ~690 instructions an entry cost)
            int x = wx0 + (int)s_q[wv][j], y = wy0;
            while (x >= W) {
                x -= W;
                ++y;
            }
            float dir[3], vdir[3], cen[3];
            ray_setup(x, y, cam, tree, dir, vdir, cen);  // only vdir is needed (rt_core.cuh:278)
            float basis_fn[RTO_BASIS_MAX_DEV];
            // (view direction + basis once per hit PIXEL, parked in LDS, removes ~200 of these ~430 instructions per entry and was
            //  built twice -- round 4 and round 6, tools/experiments/r6_shade_basis_table.patch -- and lost both times: 1.54 vs
            //  1.48 ms on C2, 1.24 vs 1.02 on C5: this arithmetic runs while the entry's record is in flight and costs nothing)
            if constexpr (MODE > 0)  // (the launcher picks MODE from the tree: SH, data_dim = MODE)
                ray_basis_sh<(MODE - 1) / 3>(opt, vdir, basis_fn);
            else if constexpr (MODE < 0)  // quantised SH tree, -MODE basis functions
                ray_basis_sh<-MODE>(opt, vdir, basis_fn);
            else
                ray_basis(tree, opt, vdir, basis_fn);
            float o[4];
            leaf_contrib<MODE>(tree, hit_slot<SPP>(he), basis_fn, (float)hit_count<SPP>(he), o);
            s_c[wv][j] = o[0];
            s_c[wv][kShadeCap + j] = o[1];
            s_c[wv][2 * kShadeCap + j] = o[2];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- pixel lanes add their entries up, in hit order
        RTO_SHADE_STAMP(3)  // (last window's) entries shaded
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int i = 0; i < SPP; ++i) {
                const uint32_t pos = start[p] + i - w0;
                if ((uint32_t)i < n[p] && pos < (uint32_t)kShadeCap) {
                    out[p][0] += s_c[wv][pos];
                    out[p][1] += s_c[wv][kShadeCap + pos];
                    out[p][2] += s_c[wv][2 * kShadeCap + pos];
                    out[p][3] += (float)hit_count<SPP>(h[p][i]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    RTO_SHADE_STAMP(4)  // sums done
    RTO_GLOBAL float* const g_aux = as_global(fd.aux);
    typedef float f4_t __attribute__((ext_vector_type(4)));  // (HIP's float4 has no assignment across address spaces)
    RTO_GLOBAL f4_t* const g_image = (RTO_GLOBAL f4_t*)as_global(fd.image);
    constexpr float INV_SPP = 1.0f / SPP;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int64_t idx = wave_px0 + p * 64 + lane;
        if (idx >= SIZE) continue;
        if (n[p]) {  // (a pixel without hits keeps its exact zeros, as before)
            out[p][0] *= INV_SPP;
            out[p][1] *= INV_SPP;
            out[p][2] *= INV_SPP;
            out[p][3] *= INV_SPP;
        }
        // volrend.cu:174-212 (write_pixel, through global-address-space pointers): background composite, then the 8 aux planes
        // and the RGBA32F image with alpha = 1 -- or, lean (block-uniform), the four values their consumers read in one store
        const float remain = opt.background_brightness * (1.f - out[p][3]);
        const float r = out[p][0] + remain, g = out[p][1] + remain, b = out[p][2] + remain, al = out[p][3];
        if (fb.lean) {
            // (sparse, level 2: nothing for a pixel of an unmarked tile -- it is the background and its consumers know it)
            if (fb.lean == 1 || ((live_bits >> p) & 1u)) g_image[idx] = f4_t{r, g, b, al};
        } else {
            RTO_GLOBAL float* a = g_aux + idx;
            a[0] = r;
            a[SIZE] = g;
            a[2 * SIZE] = b;
            a[3 * SIZE] = al;
            a[4 * SIZE] = r * r;
            a[5 * SIZE] = g * g;
            a[6 * SIZE] = b * b;
            a[7 * SIZE] = al * al;
            g_image[idx] = f4_t{r, g, b, 1.0f};
        }
    }
#ifdef RTO_DBG_COUNTERS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have left
    RTO_SHADE_STAMP(5)
    if (lane == 0 && blockIdx.x * (uint32_t)kShadeWaves + (uint32_t)wv < (uint32_t)kShadeStampWaves) {
        unsigned long long* o = g_shade_phase + (size_t)(blockIdx.x * (uint32_t)kShadeWaves + (uint32_t)wv) * 8;
        for (int i = 0; i <= 5; ++i) o[i] = ph[i];
        o[6] = total;
        o[7] = 1;
    }
#endif
}

// ------------------------------------------------------------------ frame table
// The frame descriptors of a batch reach the device as kernel arguments of this one-wave kernel, at most kFrameChunk
// per launch (a kernarg segment holds 4 KB; 64 descriptors are 6 KB), and are written to the context's table on the
// launch stream: no host staging buffer whose lifetime would have to outlast an asynchronous copy.
__global__ void write_frames_kernel(const FrameChunk c, FrameDesc* __restrict__ dst, int n) {
    const int i = threadIdx.x;
    if (i < n) dst[i] = c.f[i];
}

// ------------------------------------------------------------------ u8 conversion
// main_headless.cpp:535-538: (uint8_t)(f * 255), truncation, all four channels
__global__ void rgba8_kernel(const float4* __restrict__ in, uchar4* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = in[i];
    uchar4 o;
    o.x = (unsigned char)(v.x * 255);
    o.y = (unsigned char)(v.y * 255);
    o.z = (unsigned char)(v.z * 255);
    o.w = (unsigned char)(v.w * 255);
    out[i] = o;
}

}  // namespace rto

// ------------------------------------------------------------------ host launchers (C++ linkage,
// declared in rto_launch.h)
#include "rto_launch.h"

namespace rto {

hipError_t launch_build_nodew(const int32_t* child, const uint16_t* data, int64_t n_slots, int data_dim,
                              uint32_t* nodew, int* bad_flag, hipStream_t stream) {
    const int threads = 256;
    const int64_t blocks = (n_slots + threads - 1) / threads;
    hipLaunchKernelGGL(build_nodew_kernel, dim3((unsigned)blocks), dim3(threads), 0, stream, child, data, n_slots,
                       data_dim, nodew, bad_flag);
    return hipGetLastError();
}

hipError_t launch_build_shrec(const uint16_t* data, int64_t n_slots, int data_dim, int rec, const uint32_t* recidx, uint16_t* out,
                              hipStream_t stream) {
    const int64_t n = n_slots * rec;
    hipLaunchKernelGGL(build_shrec_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, data, n_slots, data_dim, rec, recidx, out);
    return hipGetLastError();
}

hipError_t launch_build_shrec_wide(const TreeDev& tree, const uint16_t* data, int64_t n_entries, int rec, uint16_t* out, hipStream_t stream) {
    const int64_t n = n_entries * rec;
    hipLaunchKernelGGL(build_shrec_wide_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tree, data, n_entries, rec, out);
    return hipGetLastError();
}

hipError_t launch_rebuild_reference_wide(const TreeDev& tree, int64_t n_entries, int rec, uint16_t* data, hipStream_t stream) {
    const int64_t n = n_entries * rec;
    hipLaunchKernelGGL(rebuild_reference_wide_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tree, n_entries, rec, data);
    return hipGetLastError();
}

hipError_t launch_rebuild_reference(const uint16_t* shrec, const uint32_t* nodew, const uint32_t* recidx, int64_t n_slots, int data_dim,
                                    int rec, uint16_t* data, int32_t* child, hipStream_t stream) {
    const int64_t n = n_slots * data_dim;
    hipLaunchKernelGGL(rebuild_reference_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, shrec, nodew, recidx, n_slots,
                       data_dim, rec, data, child);
    return hipGetLastError();
}

hipError_t launch_build_topgrid(const uint32_t* nodew, int G, uint2* grid, hipStream_t stream) {
    const unsigned n = 1u << (3 * G);
    hipLaunchKernelGGL(build_topgrid_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, nodew, G, grid);
    return hipGetLastError();
}

TileMap make_tile_map(int width, int height, int strip_rows) {
    TileMap tm;
    tm.tiles_x = (width + kTileW - 1) / kTileW;
    tm.tiles_y = (height + kTileH - 1) / kTileH;
    tm.strip_rows = strip_rows < 1 ? 1 : strip_rows;
    const int strips = (tm.tiles_y + tm.strip_rows - 1) / tm.strip_rows;
    const int strips_per_xcd = (strips + 7) / 8;
    tm.per_xcd = strips_per_xcd * tm.strip_rows * tm.tiles_x;
    return tm;
}

template <int SPP>
static hipError_t launch_spp(int kernel, const TreeDev& tree, const CamDev& cam, const OptDev& opt,
                             const Pcg32& rng, const PcgJumpEntry* jump, const FrameOut& fo, int strip_rows, hipStream_t stream) {
    if (kernel == 2) {
        const TileMap tm = make_tile_map(cam.width, cam.height, strip_rows);
        const size_t lds = (size_t)(tree.max_depth + 1) * 256 * sizeof(uint32_t);
        const dim3 grid(8 * tm.per_xcd), block(256);
        if (fo.stats)  // (the counting instantiation walks the one-level image: its units are defined on that walk)
            hipLaunchKernelGGL((render_fast<SPP, true, false>), grid, block, lds, stream, tree, cam, opt, rng, jump, tm, fo);
        else if (tree.widew)
            if ((tree.max_depth - tree.top_levels + 1) / 2 <= 2)  // two pairs of levels below the grid at most
                hipLaunchKernelGGL((render_fast<SPP, false, true, 1>), grid, block, lds, stream, tree, cam, opt, rng, jump, tm, fo);
            else
                hipLaunchKernelGGL((render_fast<SPP, false, true>), grid, block, lds, stream, tree, cam, opt, rng, jump, tm, fo);
        else
            hipLaunchKernelGGL((render_fast<SPP, false, false>), grid, block, lds, stream, tree, cam, opt, rng, jump, tm, fo);
    } else {
        const int64_t size = (int64_t)cam.width * cam.height;
        hipLaunchKernelGGL(render_generic<SPP>, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, stream, tree, cam,
                           opt, rng, fo);
    }
    return hipGetLastError();
}

hipError_t launch_mark_tiles_one(const TreeDev& tree, const CamDev& cam, uint32_t* mask, int mask_words, hipStream_t stream) {
    FrameDesc fd = {};
    fd.fx = cam.fx;
    fd.fy = cam.fy;
    for (int i = 0; i < 12; ++i) fd.transform[i] = cam.transform[i];
    if (hipMemsetAsync(mask, 0, (size_t)mask_words * sizeof(uint32_t), stream) != hipSuccess) return hipErrorLaunchFailure;
    if (tree.n_occ_cells > 0) {
        const dim3 grid((unsigned)((tree.n_occ_cells + kMarkCells - 1) / kMarkCells));
        const size_t lds = mask_words <= kMarkLdsWords ? (size_t)mask_words * sizeof(uint32_t) : 0;
        hipLaunchKernelGGL(mark_tiles_one_kernel, grid, dim3(256), lds, stream, tree, fd, cam.width, cam.height, mask_words, mask);
    }
    return hipGetLastError();
}

hipError_t launch_render(int kernel, int spp, const TreeDev& tree, const CamDev& cam, const OptDev& opt,
                         const Pcg32& rng, const PcgJumpEntry* jump, const FrameOut& fo, int strip_rows,
                         hipStream_t stream) {
    switch (spp) {  // volrend.cu:266-278
#ifndef RTO_DEV_SPP6_ONLY
        case 1: return launch_spp<1>(kernel, tree, cam, opt, rng, jump, fo, strip_rows, stream);
        case 2: return launch_spp<2>(kernel, tree, cam, opt, rng, jump, fo, strip_rows, stream);
        case 3: return launch_spp<3>(kernel, tree, cam, opt, rng, jump, fo, strip_rows, stream);
        case 4: return launch_spp<4>(kernel, tree, cam, opt, rng, jump, fo, strip_rows, stream);
        case 8: return launch_spp<8>(kernel, tree, cam, opt, rng, jump, fo, strip_rows, stream);
        case 16: return launch_spp<16>(kernel, tree, cam, opt, rng, jump, fo, strip_rows, stream);
        case 32: return launch_spp<32>(kernel, tree, cam, opt, rng, jump, fo, strip_rows, stream);
#endif
        case 6: return launch_spp<6>(kernel, tree, cam, opt, rng, jump, fo, strip_rows, stream);
        default: return hipErrorInvalidValue;
    }
}

// Waves per SIMD the default instantiation is built for.  Round 3: 7 (72 VGPRs).  Round 4: 8 (64 VGPRs; the spills this forces are
// in the ray set-up) -- with the shorter two-level loop one more resident wave is worth more than the spills cost: C2 4.69 -> 4.57,
// C4 13.27 -> 12.71, C5 2.96 -> 2.88 ms per 100 frames in one box (profiles/r4_u_ab_wps8.txt).  Tuning key refill = 732: the 7-wave build.
#ifndef RTO_WPS_DEFAULT
#define RTO_WPS_DEFAULT 8
#endif
template <int SPP, int REFILL, int WPS, bool WIDE>
static hipError_t launch_batch_impl(const TreeDev& tree, const OptDev& opt, const FrameBatch& fb,
                                    const PcgJumpEntry* jump, unsigned long long* queue, uint32_t* hits, int num_cus,
                                    int chunk_override, bool cull, OccupancyCache* occ, hipEvent_t* ev, hipStream_t stream) {
    // dynamic LDS: ancestor stack + thresholds per lane, then the frame table of THIS batch (96 B per frame: a batch of
    // one does not pay for 128)
    // (two pairs of levels below the grid at most: the ancestor stack is two registers and its LDS rows only park a ray's two
    //  hand-off offsets)
    const bool regstack = WIDE && (tree.max_depth - tree.top_levels + 1) / 2 <= 2;
    const size_t lds = (size_t)((regstack ? 2 : tree.max_depth + 1 - tree.top_levels) + SPP + 1) * 256 * sizeof(uint32_t) + sizeof(float) * kCamFloats * (size_t)fb.n;
    const auto kern = regstack ? &render_persist<SPP, REFILL, WPS, WIDE, WIDE ? 1 : 0> : &render_persist<SPP, REFILL, WPS, WIDE, 0>;
    const void* fn = reinterpret_cast<const void*>(kern);
    OccupancyCache local;
    if (!occ) occ = &local;
    occ->lds_refused = false;
    if (occ->force_lds_refusal) {
        occ->lds_refused = true;
        return hipSuccess;
    }
    if (occ->blocks_per_cu == 0 || occ->fn != fn || occ->lds != lds) {
        if (lds > 64 * 1024) {  // beyond the default dynamic-LDS window: ask for it (the CU has 160 KB)
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                (void)hipGetLastError();
                occ->lds_refused = true;  // (deep tree x SPP 32 x many frames) nothing launched: the caller takes the generic kernel
                return hipSuccess;
            }
        }
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1)
            nb = 2;
        occ->blocks_per_cu = nb > 8 ? 8 : nb;
        occ->fn = fn;
        occ->lds = lds;
    }
    const int blocks_per_cu = occ->cap > 0 && occ->cap < occ->blocks_per_cu ? occ->cap : occ->blocks_per_cu;
    const int tiles = ((fb.width + 7) / 8) * ((fb.height + 7) / 8) * fb.n;
    int grid = num_cus * blocks_per_cu;
    if (grid > (tiles + 3) / 4) grid = (tiles + 3) / 4;  // never more waves than 8x8 tiles
    if (grid < 1) grid = 1;
    // dequeue granularity: big enough to keep the single counter far below its ~90 dequeues/us,
    // small enough that every wave draws several times (a wave that draws twice while its
    // neighbour draws three times is a 33 % imbalance)
    const int64_t rays_per_wave = (int64_t)tiles * 64 / ((int64_t)grid * 4);
    // (round 6, with the band queues: two tiles per dequeue beat four on every configuration -- C2 3.83 -> 3.74 ms per 100 frames,
    //  C5 2.44 -> 2.41, C4 11.85 -> 11.71, 8 scenes 3.26 -> 3.13; one tile 3.74 / 2.48 / 11.73 / 3.14, eight 3.90 on C2:
    //  profiles/r6_zz_ab_chunk_*.txt)
    const uint32_t chunk = chunk_override > 0 ? (uint32_t)chunk_override : (rays_per_wave >= 512 ? 128u : 64u);
    const int64_t size = (int64_t)fb.width * fb.height;
    if (ev) (void)hipEventRecord(ev[0], stream);
    // tile marks (empty-space culling; all ones when it is off), then the ray queues as lists of the marked tile slots
    uint32_t* mask = const_cast<uint32_t*>(fb.tile_mask);
    const size_t mask_bytes = (size_t)fb.n * fb.mask_words * sizeof(uint32_t);
    if (hipMemsetAsync(mask, cull ? 0 : 0xff, mask_bytes, stream) != hipSuccess) return hipErrorLaunchFailure;
    if (cull && tree.n_occ_cells > 0) {
        const dim3 mgrid((unsigned)((tree.n_occ_cells + kMarkCells - 1) / kMarkCells), fb.n);
        if (fb.mask_words <= kMarkLdsWords)
            hipLaunchKernelGGL(mark_tiles_kernel<true>, mgrid, dim3(256), (size_t)fb.mask_words * sizeof(uint32_t), stream, tree, fb, mask);
        else
            hipLaunchKernelGGL(mark_tiles_kernel<false>, mgrid, dim3(256), 0, stream, tree, fb, mask);
    }
    const unsigned n_chunks = (unsigned)fb.qchunk[fb.n_queues];
    hipLaunchKernelGGL(queue_count_kernel, dim3(n_chunks), dim3(kQueueChunk), 0, stream, fb);
    hipLaunchKernelGGL(queue_scan_kernel, dim3(1), dim3(64 * kMaxQueues), 0, stream, fb, queue);  // (+ arms the ray queues)
    hipLaunchKernelGGL(queue_write_kernel, dim3(n_chunks), dim3(kQueueChunk), 0, stream, fb);
    hipLaunchKernelGGL(sample_kernel<SPP>, dim3((unsigned)(((tiles / fb.n + kSampleTiles - 1) / kSampleTiles + kSampleWaves - 1) / kSampleWaves), fb.n), dim3(64 * kSampleWaves), 0, stream, fb, jump);
    if (ev) (void)hipEventRecord(ev[1], stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, tree, opt, fb, queue, hits, chunk);
    if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
    if (ev) (void)hipEventRecord(ev[2], stream);
#ifndef RTO_SHADE_P
#define RTO_SHADE_P 2
#endif
    constexpr int SP = SPP <= 8 ? RTO_SHADE_P : 1;  // pixels per lane of the shading kernel (its hit lists live in registers)
    const unsigned pblocks = (unsigned)((size + 64 * kShadeWaves * SP - 1) / (64 * kShadeWaves * SP));
#if RTO_SHADE_BAND_ROWS > 0
    const unsigned cpb = shade_blocks_per_band(fb.width, 64 * kShadeWaves * SP), bands = (pblocks + cpb - 1u) / cpb;
    const dim3 sgrid(((bands + 7u) / 8u) * 8u * cpb * (unsigned)fb.n);  // see shade_kernel: (band, frame, pixel block of the band) <- block id
#else
    const dim3 sgrid(((pblocks + 7u) / 8u) * 8u * (unsigned)fb.n);  // see shade_kernel: (pixel block, frame) <- block id
#endif
#define RTO_SHADE(M) hipLaunchKernelGGL((shade_kernel<SPP, SP, M>), sgrid, dim3(64 * kShadeWaves), 0, stream, tree, opt, fb, (const uint32_t*)hits)
    if (tree.qrec) {  // (the host admits SH4/9/16/25 only)
        if (tree.basis_dim == 4)
            RTO_SHADE(-4);
        else if (tree.basis_dim == 9)
            RTO_SHADE(-9);
        else if (tree.basis_dim == 16)
            RTO_SHADE(-16);
        else
            RTO_SHADE(-25);
    } else if (tree.format == 1 && tree.data_dim == 28)
        RTO_SHADE(28);
    else if (tree.format == 1 && tree.data_dim == 49)
        RTO_SHADE(49);
    else if (tree.format == 1 && tree.data_dim == 76)
        RTO_SHADE(76);
    else
        RTO_SHADE(0);
#undef RTO_SHADE
    if (ev) (void)hipEventRecord(ev[3], stream);
    return hipGetLastError();
}

template <int SPP>
static hipError_t launch_batch_spp(const TreeDev& tree, const OptDev& opt, const FrameBatch& fb,
                                   const PcgJumpEntry* jump, unsigned long long* queue, uint32_t* hits, int num_cus,
                                   int refill, bool cull, OccupancyCache* occ, hipEvent_t* ev, hipStream_t stream) {
    // tuning: refill = 1000 * tiles_per_dequeue + 100 * waves/SIMD + threshold
    refill %= 100000;
    const int chunk_override = (refill / 1000) * 64;
    refill %= 1000;
    const bool wide = tree.widew != nullptr;
    if constexpr (SPP == 6) {  // tuning instantiations only for the benchmark configuration (and its usual two-level image)
#define RTO_F(R, O) return launch_batch_impl<SPP, R, O, true>(tree, opt, fb, jump, queue, hits, num_cus, chunk_override, cull, occ, ev, stream)
        if (wide) switch (refill) {  // A/B set kept for tools/ab_tuning.py: 100 * waves/SIMD + refill threshold
            case 808: RTO_F(8, 8);
            case 816: RTO_F(16, 8);
            case 824: RTO_F(24, 8);
            case 832: RTO_F(32, 8);
            case 840: RTO_F(40, 8);
            case 724: RTO_F(24, 7);
            case 732: RTO_F(32, 7);
            case 740: RTO_F(40, 7);
            case 632: RTO_F(32, 6);
            case 432: RTO_F(32, 4);
            case 232: RTO_F(32, 2);
            case 132: RTO_F(32, 1);
            default: break;
        }
#undef RTO_F
    }
    // Refill once half the lanes are idle (larger refill rounds waste fewer issue slots on the partially filled ray set-up:
    // 32 idle lanes beat 16 by 4 %); registers budgeted for RTO_WPS_DEFAULT waves per SIMD.  Occupancy matters (round 3,
    // measured with the real knob, tuning key blocks_per_cu: 1 / 2 / 3 / 4 / 5 / 6 workgroups per CU take 26.2 / 14.4 /
    // 10.6 / 8.8 / 7.8 / 7.35 ms per 100 frames -- round 2's "4 to 8 waves within 2 %" compared __launch_bounds__ hints,
    // which change the register budget, not the number of resident waves).
    // The two-level traversal image when the tree has one (always, unless it would not fit its index space or the device's
    // memory: rto_abi.cpp build_wide_image), else the one-level image: the same pixels either way.
    // (Round 5's reservoir kernel -- whole-tile set-up, rays parked in LDS, refill rounds at 8-24 idle lanes -- lost its same-box
    //  A/B, 4.27-4.32 against 4.11-4.20 ms per 100 C2 frames, and lives in tools/experiments/r5_lab_switches.patch.)
    if (wide)
        return launch_batch_impl<SPP, 32, RTO_WPS_DEFAULT, true>(tree, opt, fb, jump, queue, hits, num_cus, chunk_override, cull, occ, ev, stream);
    return launch_batch_impl<SPP, 32, RTO_WPS_DEFAULT, false>(tree, opt, fb, jump, queue, hits, num_cus, chunk_override, cull, occ, ev, stream);
}

hipError_t launch_render_batch(int spp, const TreeDev& tree, const OptDev& opt, const FrameBatch& fb,
                               const PcgJumpEntry* jump, unsigned long long* queue, uint32_t* hits, int num_cus,
                               int refill, bool cull, OccupancyCache* occ, hipEvent_t* ev, hipStream_t stream) {
    switch (spp) {
#ifndef RTO_DEV_SPP6_ONLY  // (development builds: compile the benchmark's instantiation only)
        case 1: return launch_batch_spp<1>(tree, opt, fb, jump, queue, hits, num_cus, refill, cull, occ, ev, stream);
        case 2: return launch_batch_spp<2>(tree, opt, fb, jump, queue, hits, num_cus, refill, cull, occ, ev, stream);
        case 3: return launch_batch_spp<3>(tree, opt, fb, jump, queue, hits, num_cus, refill, cull, occ, ev, stream);
        case 4: return launch_batch_spp<4>(tree, opt, fb, jump, queue, hits, num_cus, refill, cull, occ, ev, stream);
#endif
        case 6: return launch_batch_spp<6>(tree, opt, fb, jump, queue, hits, num_cus, refill, cull, occ, ev, stream);
#ifndef RTO_DEV_SPP6_ONLY
        case 8: return launch_batch_spp<8>(tree, opt, fb, jump, queue, hits, num_cus, refill, cull, occ, ev, stream);
        case 16: return launch_batch_spp<16>(tree, opt, fb, jump, queue, hits, num_cus, refill, cull, occ, ev, stream);
        case 32: return launch_batch_spp<32>(tree, opt, fb, jump, queue, hits, num_cus, refill, cull, occ, ev, stream);
#endif
        default: return hipErrorInvalidValue;
    }
}

#ifdef RTO_DBG_COUNTERS
hipError_t debug_shade_phases(unsigned long long* out, bool reset) {  // out: kShadeStampWaves * 8 words
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_shade_phase), sizeof(unsigned long long) * kShadeStampWaves * 8);
    if (e == hipSuccess && reset) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_shade_phase));
        if (e == hipSuccess) e = hipMemset(p, 0, sizeof(unsigned long long) * kShadeStampWaves * 8);
    }
    return e;
}
#endif

hipError_t launch_write_frames(const FrameDesc* host, int n, FrameDesc* dev_table, hipStream_t stream) {
    for (int f0 = 0; f0 < n; f0 += kFrameChunk) {
        FrameChunk c;
        const int m = n - f0 < kFrameChunk ? n - f0 : kFrameChunk;
        for (int i = 0; i < m; ++i) c.f[i] = host[f0 + i];
        hipLaunchKernelGGL(write_frames_kernel, dim3(1), dim3(64), 0, stream, c, dev_table + f0, m);
    }
    return hipGetLastError();
}

hipError_t launch_rgba8(const float* rgba, uint8_t* out, int64_t n_pixels, hipStream_t stream) {
    const int threads = 256;
    hipLaunchKernelGGL(rgba8_kernel, dim3((unsigned)((n_pixels + threads - 1) / threads)), dim3(threads), 0, stream,
                       reinterpret_cast<const float4*>(rgba), reinterpret_cast<uchar4*>(out), n_pixels);
    return hipGetLastError();
}

}  // namespace rto
