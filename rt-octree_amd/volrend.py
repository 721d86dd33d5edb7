"""Host-side mirror of the reference's render operator interface, over the C ABI (include/rto.h).

Same names, argument meaning and error behaviour as the reference (paths relative to
/root/reference):
    RenderOptions    renderer/include/volrend/render_options.hpp:13-78
    N3Tree           renderer/include/volrend/n3tree.hpp, src/n3tree.cpp:111-362
    Camera           renderer/include/volrend/camera.hpp (transform = glm::mat4x3, column-major)
    RenderContext    renderer/include/volrend/render_context.hpp:14-214 (rng, aux, images, Timer)
    launch_renderer  renderer/include/volrend/cuda/renderer_kernel.hpp:11-16
    filtering        denoiser/extension/filtering.h:7-13

All pixel work runs in the HIP kernels behind librto.so; this module only marshals arguments.
"""
import ctypes as C
import json
import os

import numpy as np

from . import _lib
from ._lib import CCamera, COptions, CTreeInfo, RtoError, check, lib

SUPPORTED_SPP = (1, 2, 3, 4, 6, 8, 16, 32)  # volrend.cu:266-278
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_FAST = 0, 1, 2
AUX_CHANNELS = 8  # render_context.hpp:23
_FORMAT_NAMES = {0: "RGBA", 1: "SH", 2: "SG", 3: "ASG"}


def _stream_ptr(stream):
    """None -> default stream; int -> raw hipStream_t; torch.cuda.Stream -> its handle."""
    if stream is None:
        return C.c_void_p(0)
    if isinstance(stream, int):
        return C.c_void_p(stream)
    if hasattr(stream, "cuda_stream"):
        return C.c_void_p(stream.cuda_stream)
    raise TypeError("stream must be None, an int handle or a torch.cuda.Stream")


class RenderOptions:
    """render_options.hpp:13-78.  Field names and defaults are the reference's."""

    _JSON_KEYS = ("step_size", "sigma_thresh", "stop_thresh", "background_brightness", "show_grid",
                  "grid_max_depth", "enable_probe", "probe", "probe_disp_size", "denoise", "spp")
    SPP_DEFAULT = 4  # render_options.hpp:57

    def __init__(self, **kw):
        c = COptions()
        lib().rto_options_default(C.byref(c))
        self._load(c)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError("RenderOptions has no field '%s'" % k)
            setattr(self, k, v)

    def _load(self, c):
        self.step_size = c.step_size
        self.sigma_thresh = c.sigma_thresh
        self.stop_thresh = c.stop_thresh
        self.background_brightness = c.background_brightness
        self.render_bbox = list(c.render_bbox)
        self.basis_minmax = list(c.basis_minmax)
        self.rot_dirs = list(c.rot_dirs)
        self.show_grid = bool(c.show_grid)
        self.grid_max_depth = c.grid_max_depth
        self.render_depth = bool(c.render_depth)
        self.enable_probe = bool(c.enable_probe)
        self.probe = list(c.probe)
        self.probe_disp_size = c.probe_disp_size
        self.denoise = bool(c.denoise)
        self.spp = c.spp

    @classmethod
    def from_json(cls, path):
        """`options = json::parse(f)` (main_headless.cpp:459-464): all 11 keys are required."""
        c = COptions()
        check(lib().rto_options_from_json_file(os.fsencode(path), C.byref(c)))
        o = cls.__new__(cls)
        o._load(c)
        return o

    @classmethod
    def from_json_text(cls, text):
        c = COptions()
        check(lib().rto_options_from_json(text.encode("utf-8"), C.byref(c)))
        o = cls.__new__(cls)
        o._load(c)
        return o

    def to_json(self):
        return json.dumps({k: getattr(self, k) for k in self._JSON_KEYS}, indent=2, sort_keys=True)

    def to_c(self):
        c = COptions()
        c.step_size, c.sigma_thresh, c.stop_thresh = self.step_size, self.sigma_thresh, self.stop_thresh
        c.background_brightness = self.background_brightness
        for i in range(6):
            c.render_bbox[i] = self.render_bbox[i]
        for i in range(2):
            c.basis_minmax[i] = int(self.basis_minmax[i])
        for i in range(3):
            c.rot_dirs[i] = self.rot_dirs[i]
            c.probe[i] = self.probe[i]
        c.show_grid, c.grid_max_depth = int(self.show_grid), int(self.grid_max_depth)
        c.render_depth, c.enable_probe = int(self.render_depth), int(self.enable_probe)
        c.probe_disp_size = int(self.probe_disp_size)
        c.denoise, c.spp = int(self.denoise), int(self.spp)
        return c


class N3Tree:
    """Device-resident PlenOctree.  `N3Tree(path)` = N3Tree::open + load_cuda
    (n3tree.cpp:111-154, n3tree.cu:9-41)."""

    def __init__(self, path=None, device=0, quant_direct=False, compact=False, keep_reference=False, compact_records=False,
                 no_culling=False):
        self._h = C.c_void_p(0)
        self.device = device
        self.quant_direct = bool(quant_direct)  # render a quantised tree from its codebooks (no expansion)
        self.compact = bool(compact)            # RTO_TREE_COMPACT: no aligned copy of the SH coefficients for shading
        self.keep_reference = bool(keep_reference)  # RTO_TREE_KEEP_REFERENCE: child[] / data[] stay resident
        self.compact_records = bool(compact_records)  # RTO_TREE_COMPACT_RECORDS: coefficient records for hittable leaves only
        self.no_culling = bool(no_culling)  # RTO_TREE_NO_CULLING: no empty-space culling cells
        if path is not None:
            self.open(path)

    def _flags(self):
        return ((1 if self.quant_direct else 0) | (2 if self.compact else 0) | (4 if self.keep_reference else 0)
                | (8 if self.compact_records else 0) | (16 if self.no_culling else 0))

    def open(self, path):
        self.free()
        h = C.c_void_p(0)
        check(lib().rto_tree_load_npz_ex(os.fsencode(path), self.device, self._flags(), C.byref(h)))
        self._h = h
        self._refresh()

    @classmethod
    def from_arrays(cls, child, data, scale, offset, data_format="", device=0, compact=False, keep_reference=False,
                    compact_records=False, no_culling=False):
        """child int32 [capacity,N,N,N]; data float16 (or uint16 bits) [capacity,N,N,N,data_dim];
        scale = invradius3, offset (n3tree.cpp:257-267)."""
        child = np.ascontiguousarray(child, dtype=np.int32)
        data = np.ascontiguousarray(data)
        if data.dtype == np.float16:
            data = data.view(np.uint16)
        if data.dtype != np.uint16:
            raise RtoError(-6, "data must be stored in half precision")  # n3tree.cpp:345
        if child.ndim != 4 or data.ndim != 5:
            raise RtoError(-1, "child must be [capacity,N,N,N] and data [capacity,N,N,N,data_dim]")
        cap, N, dd = child.shape[0], child.shape[1], data.shape[-1]
        sc = (C.c_float * 3)(*[float(x) for x in np.broadcast_to(np.asarray(scale, np.float32), (3,))])
        of = (C.c_float * 3)(*[float(x) for x in np.broadcast_to(np.asarray(offset, np.float32), (3,))])
        t = cls(device=device, compact=compact, keep_reference=keep_reference, compact_records=compact_records,
                no_culling=no_culling)
        h = C.c_void_p(0)
        check(lib().rto_tree_from_arrays_ex(C.c_void_p(child.ctypes.data), C.c_void_p(data.ctypes.data), cap, N, dd,
                                            data_format.encode("ascii"), sc, of, device, t._flags(), C.byref(h)))
        t._h = h
        t._refresh()
        return t

    def _refresh(self):
        info = CTreeInfo()
        check(lib().rto_tree_get_info(self._h, C.byref(info)))
        self.capacity, self.N, self.data_dim = info.capacity, info.N, info.data_dim
        self.basis_dim = info.basis_dim
        self.data_format = _FORMAT_NAMES.get(info.format, "UNKNOWN") + (str(info.basis_dim) if info.basis_dim != -1 else "")
        self.scale = np.array(info.scale, np.float32)
        self.offset = np.array(info.offset, np.float32)
        self.use_ndc = bool(info.use_ndc)
        self.ndc_width, self.ndc_height, self.ndc_focal = info.ndc_width, info.ndc_height, info.ndc_focal
        self.max_depth = info.max_depth
        self.device_bytes = info.device_bytes
        self.wide_nodes = info.wide_nodes

    def set_ndc(self, width, height, focal):
        """main_headless.cpp:400-405: tree.use_ndc = true; ndc_width/height/focal."""
        check(lib().rto_tree_set_ndc(self._h, float(width), float(height), float(focal)))
        self._refresh()

    def is_data_loaded(self):
        return bool(self._h)

    def free(self):
        if getattr(self, "_h", None):
            lib().rto_tree_free(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Camera:
    """camera.hpp: width, height, fx, fy and `transform`, a glm::mat4x3 (4 columns of 3): columns
    0..2 are the camera axes, column 3 the centre.  `transform` here is a float32 [4,3] array whose
    row c is glm column c, so `.reshape(-1)` is the 12-float column-major block the kernel reads
    (common.cuh:29-44)."""

    def __init__(self, width=800, height=800, fx=1111.11, fy=-1.0):
        self.width, self.height = int(width), int(height)
        self.fx = float(fx)
        self.fy = float(fy) if fy > 0 else float(fx)
        self.transform = np.zeros((4, 3), np.float32)
        self.transform[0, 0] = self.transform[1, 1] = self.transform[2, 2] = 1.0

    def set_c2w(self, m):
        """m: row-major 3x4 / 4x4 camera-to-world (blender `transform_matrix`); transposed into the
        column-major layout like main_headless.cpp:262-268."""
        m = np.asarray(m, np.float32)
        self.transform = np.ascontiguousarray(m[:3, :4].T)

    def to_c(self):
        c = CCamera()
        c.width, c.height, c.fx, c.fy = self.width, self.height, self.fx, self.fy
        flat = np.asarray(self.transform, np.float32).reshape(-1)
        if flat.size != 12:
            raise RtoError(-1, "Camera.transform must hold 12 floats")
        for i in range(12):
            c.transform[i] = float(flat[i])
        return c


class _DevArray:
    """Zero-copy view of a device buffer for torch.as_tensor / cupy (CUDA array interface v2)."""

    def __init__(self, ptr, shape, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner


class Timer:
    """RenderContext::Timer (render_context.hpp:122-213): three event pairs on the render stream,
    FPS = 1000 / (render + torch + filter) over the recorded frames."""

    RENDER, TORCH, FILTER = 0, 1, 2

    def __init__(self, ctx):
        self._ctx = ctx

    def reset(self, stream=None):
        check(lib().rto_timer_reset(self._ctx._h, _stream_ptr(stream)))

    def render_start(self): check(lib().rto_timer_start(self._ctx._h, 0))
    def render_stop(self): check(lib().rto_timer_stop(self._ctx._h, 0))
    def torch_start(self): check(lib().rto_timer_start(self._ctx._h, 1))
    def torch_stop(self): check(lib().rto_timer_stop(self._ctx._h, 1))
    def filter_start(self): check(lib().rto_timer_start(self._ctx._h, 2))
    def filter_stop(self): check(lib().rto_timer_stop(self._ctx._h, 2))

    def record(self, denoise):
        check(lib().rto_timer_record(self._ctx._h, int(bool(denoise))))

    def stats(self):
        ms = (C.c_float * 3)()
        fps, n = C.c_float(0), C.c_int(0)
        check(lib().rto_timer_report(self._ctx._h, ms, C.byref(fps), C.byref(n)))
        return {"render_ms": ms[0], "torch_ms": ms[1], "filter_ms": ms[2],
                "all_ms": ms[0] + ms[1] + ms[2], "fps": fps.value, "frames": n.value}

    def report(self):
        s = self.stats()
        print("render: %.10f ms per frame" % s["render_ms"])
        print("torch:  %.10f ms per frame" % s["torch_ms"])
        print("filter: %.10f ms per frame" % s["filter_ms"])
        print("all:    %.10f ms per frame" % s["all_ms"])
        print("FPS:    %.10f" % s["fps"])
        return s


class RenderContext:
    """render_context.hpp:14-214 with offscreen = true: rng = pcg32(20230418), aux_buffer
    [8,H,W] f32, noisy image and final image [H,W,4] f32 (linear device memory instead of
    cudaArray + surface/texture objects)."""

    CHANNELS = AUX_CHANNELS

    def __init__(self, width, height, device=0, frames=1):
        self._h = C.c_void_p(0)
        h = C.c_void_p(0)
        check(lib().rto_ctx_create_batch(int(width), int(height), int(frames), int(device), C.byref(h)))
        self._h = h
        self.width, self.height, self.device, self.frames = int(width), int(height), int(device), int(frames)
        self.offscreen = True
        self._timer = Timer(self)

    # rng (pcg32.h)
    def rng_seed(self, initstate=20230418, initseq=1):
        lib().rto_ctx_rng_seed(self._h, initstate, initseq)

    def rng_advance(self, delta=1 << 32):
        """ctx.rng.advance() (main_headless.cpp:478,506): default jump 2^32."""
        lib().rto_ctx_rng_advance(self._h, delta)

    def rng_set(self, state, inc):
        lib().rto_ctx_rng_set(self._h, state, inc)

    def rng_get(self):
        s, i = C.c_uint64(0), C.c_uint64(0)
        lib().rto_ctx_rng_get(self._h, C.byref(s), C.byref(i))
        return s.value, i.value

    def select_frame(self, frame):
        """frame slot the single-frame entry points, accessors and downloads refer to"""
        check(lib().rto_ctx_select_frame(self._h, int(frame)))

    def batch_views(self):
        """zero-copy views over ALL frame slots: aux [F,8,H,W], noisy [F,H,W,4], image [F,H,W,4]"""
        sel = lib().rto_ctx_selected_frame(self._h)  # (the selection is the caller's: left as found)
        self.select_frame(0)
        F, H, W = self.frames, self.height, self.width
        views = (_DevArray(self.aux_ptr, (F, AUX_CHANNELS, H, W), self), _DevArray(self.noisy_ptr, (F, H, W, 4), self),
                 _DevArray(self.image_ptr, (F, H, W, 4), self))
        self.select_frame(sel)
        return views

    def set_kernel(self, kernel):
        check(lib().rto_ctx_set_kernel(self._h, int(kernel)))

    def timer(self):
        return self._timer

    def set_tuning(self, key, value):
        """performance knobs ("strip_rows", "refill", "tile_order", "xcd_queues", "tile_major", "tile_block", "blocks_per_cu", "cull", "cull_single");
        results never change"""
        check(lib().rto_ctx_set_tuning(self._h, key.encode("ascii"), int(value)))

    def set_lean_outputs(self, on=True):
        """rto_ctx_set_lean_outputs: batched launches with denoise on store the noisy image as (r, g, b, alpha) and no aux
        planes -- 16 instead of 48 bytes per pixel; consumers: FusedGuidanceNet(..., rgba=True) on noisy_ptr, denoise().
        on = 2 (sparse): and nothing at all for the pixels of culled tiles; consumers additionally pass sparse=True + the marks"""
        check(lib().rto_ctx_set_lean_outputs(self._h, int(on)))

    def frames_lean_level(self, first=0, n=1):
        return lib().rto_ctx_frames_lean_level(self._h, int(first), int(n))  # 0 full / 1 lean / 2 sparse; -1: a mixed range

    def frames_are_lean(self, first=0, n=1):
        return lib().rto_ctx_frames_are_lean(self._h, int(first), int(n)) == 1  # (0: all full; -1: a mixed range)

    def kernel_timing(self, on=True):
        """HIP-event timing of the traversal / shading kernels of launch_renderer_batch"""
        check(lib().rto_ctx_kernel_timing(self._h, int(bool(on))))

    def kernel_timing_read(self):
        g, t, s, n = C.c_float(0), C.c_float(0), C.c_float(0), C.c_int(0)
        check(lib().rto_ctx_kernel_timing_read3(self._h, C.byref(g), C.byref(t), C.byref(s), C.byref(n)))
        return {"raygen_ms": g.value, "traverse_ms": t.value, "shade_ms": s.value, "launches": n.value}

    def queue_stats(self):
        """(tile slots marched, tile slots in all) of the last batched launch: the rest were culled as provably empty"""
        a, b = C.c_int64(0), C.c_int64(0)
        check(lib().rto_ctx_queue_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def tile_marks(self):
        """(device pointer, words per frame, first slot, frames, background) of the tile marks the last batched launch left
        (rto_ctx_tile_marks) -- what FusedGuidanceNet.filter_packed(cull=...) takes; None after a single-frame launch"""
        p, w, s0, n, bg = C.c_void_p(None), C.c_int(0), C.c_int(0), C.c_int(0), C.c_float(0)
        if lib().rto_ctx_tile_marks(self._h, C.byref(p), C.byref(w), C.byref(s0), C.byref(n), C.byref(bg)) != 0:
            return None
        return p.value, w.value, s0.value, n.value, bg.value

    def enable_stats(self, on=True, marched=False):
        """Work counters for the roofline's algorithmic byte count (never in a timed run).  marched: also count the frame as
        the batched path works through it (get_march_stats) -- select a slot of the last batched launch and re-render its pose."""
        check(lib().rto_ctx_enable_stats(self._h, 2 if (on and marched) else int(bool(on))))

    def get_march_stats(self, reset=True, stream=None):
        out = (C.c_uint64 * 8)()
        check(lib().rto_ctx_get_march_stats(self._h, _stream_ptr(stream), out, int(bool(reset))))
        keys = ("rays", "steps", "grid_loads", "node_loads", "hit_entries", "rays_in_box", "wide_loads")
        return {k: int(out[i]) for i, k in enumerate(keys)}

    def get_stats(self, reset=True, stream=None):
        out = (C.c_uint64 * 6)()
        check(lib().rto_ctx_get_stats(self._h, _stream_ptr(stream), out, int(bool(reset))))
        keys = ("rays", "rays_in_box", "steps", "levels", "hit_leaves", "hit_rays")
        return {k: int(v) for k, v in zip(keys, out)}

    # device pointers / zero-copy views
    @property
    def aux_ptr(self): return lib().rto_ctx_aux(self._h)
    @property
    def noisy_ptr(self): return lib().rto_ctx_noisy(self._h)
    @property
    def image_ptr(self): return lib().rto_ctx_image(self._h)

    def aux_view(self): return _DevArray(self.aux_ptr, (1, AUX_CHANNELS, self.height, self.width), self)
    def noisy_view(self): return _DevArray(self.noisy_ptr, (self.height, self.width, 4), self)
    def image_view(self): return _DevArray(self.image_ptr, (self.height, self.width, 4), self)

    # host copies (main_headless.cpp:508-540)
    def download_aux(self, stream=None):
        out = np.empty((AUX_CHANNELS, self.height, self.width), np.float32)
        check(lib().rto_ctx_download_aux(self._h, _stream_ptr(stream), C.c_void_p(out.ctypes.data)))
        return out

    def download_image(self, noisy=False, stream=None):
        out = np.empty((self.height, self.width, 4), np.float32)
        check(lib().rto_ctx_download_image(self._h, _stream_ptr(stream), int(noisy), C.c_void_p(out.ctypes.data)))
        return out

    def download_rgba8(self, noisy=False, stream=None):
        out = np.empty((self.height, self.width, 4), np.uint8)
        check(lib().rto_ctx_download_rgba8(self._h, _stream_ptr(stream), int(noisy), C.c_void_p(out.ctypes.data)))
        return out

    def freeResource(self):
        if getattr(self, "_h", None):
            lib().rto_ctx_free(self._h)
            self._h = C.c_void_p(0)

    free = freeResource

    def __del__(self):
        try:
            self.freeResource()
        except Exception:
            pass


def launch_renderer(tree, cam, options, ctx, stream=None, offscreen=True):
    """volrend::launch_renderer(tree, cam, options, ctx, stream, offscreen)
    (renderer_kernel.hpp:11-16).  Asynchronous on `stream`.  Unsupported spp raises like the
    reference's std::runtime_error("spp == N not supported.") (volrend.cu:275-277)."""
    if not offscreen:
        raise RtoError(-3, "only the offscreen (headless) path is built; GL interop is out of scope")
    cc, co = cam.to_c(), options.to_c()
    check(lib().rto_launch_renderer(tree._h, C.byref(cc), C.byref(co), ctx._h, _stream_ptr(stream)))


def launch_renderer_batch(tree, cams, options, ctx, stream=None, rng_jumps=None):
    """n frames in one launch of the persistent ray-queue kernel (rto_launch_renderer_batch):
    cams[f] -> frame slot f, RNG = ctx.rng advanced by rng_jumps[f] (default f) jumps of 2^32.
    Bit-identical to the reference's frame loop `launch_renderer(...); ctx.rng.advance()`."""
    n = len(cams)
    arr = (CCamera * n)(*[c.to_c() for c in cams])
    jumps = None
    if rng_jumps is not None:
        jumps = (C.c_int64 * n)(*[int(j) for j in rng_jumps])
    co = options.to_c()
    check(lib().rto_launch_renderer_batch(tree._h, arr, jumps, n, C.byref(co), ctx._h, _stream_ptr(stream)))


def _dev_ptr(t):
    if isinstance(t, int):
        return C.c_void_p(t)
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    if hasattr(t, "__cuda_array_interface__"):
        return C.c_void_p(t.__cuda_array_interface__["data"][0])
    raise TypeError("expected a device tensor or pointer")


FILTER_EXACT, FILTER_FAST = 0, 1  # RTO_FILTER_EXACT / RTO_FILTER_FACTORISED (include/rto.h)


def filtering(stream, weight_map, guidance_map, img_in, img_out, mode=FILTER_EXACT):
    """denoiser::filtering(stream, weight_map[L,H,W], guidance_map[L,H,W], img_in, img_out)
    (filtering.h:7-13).  Tensors are contiguous float32 device tensors (torch) or raw pointers with
    `shape`; img_in / img_out are [H,W,4].  mode: FILTER_EXACT (bit-identical to the oracle) or
    FILTER_FAST (factorised exponentials, ~1e-6 relative)."""
    for t in (weight_map, guidance_map):
        if hasattr(t, "is_contiguous") and not t.is_contiguous():
            raise RtoError(-1, "weight_map / guidance_map must be contiguous")  # CHECK_CONTIGUOUS
    L, H, W = (int(s) for s in guidance_map.shape[-3:])
    n = int(guidance_map.shape[0]) if len(guidance_map.shape) == 4 else 1  # [n,L,H,W]: n images per launch
    check(lib().rto_filtering_batch_mode(_stream_ptr(stream), _dev_ptr(weight_map), _dev_ptr(guidance_map), L, H, W, n,
                                         _dev_ptr(img_in), _dev_ptr(img_out), int(mode)))
