"""Loads librto.so (the C ABI of include/rto.h) with ctypes.  No fallback: if the HIP library is
missing or does not load, importing a symbol from it raises."""
import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("RTO_LIB") or os.path.join(PKG_DIR, "lib", "librto.so")  # (RTO_LIB: another build of the same library, for same-box A/B runs)

RTO_OK = 0


class RtoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rto error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


class COptions(C.Structure):
    """rto_options (include/rto.h) == RenderOptions (render_options.hpp:13-78)."""
    _fields_ = [
        ("step_size", C.c_float), ("sigma_thresh", C.c_float), ("stop_thresh", C.c_float),
        ("background_brightness", C.c_float), ("render_bbox", C.c_float * 6),
        ("basis_minmax", C.c_int * 2), ("rot_dirs", C.c_float * 3),
        ("show_grid", C.c_int), ("grid_max_depth", C.c_int), ("render_depth", C.c_int),
        ("enable_probe", C.c_int), ("probe", C.c_float * 3), ("probe_disp_size", C.c_int),
        ("denoise", C.c_int), ("spp", C.c_int),
    ]


class CCamera(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fx", C.c_float), ("fy", C.c_float),
                ("transform", C.c_float * 12)]


class CTreeInfo(C.Structure):
    _fields_ = [
        ("capacity", C.c_int64), ("N", C.c_int), ("data_dim", C.c_int), ("format", C.c_int),
        ("basis_dim", C.c_int), ("scale", C.c_float * 3), ("offset", C.c_float * 3),
        ("use_ndc", C.c_int), ("ndc_width", C.c_float), ("ndc_height", C.c_float),
        ("ndc_focal", C.c_float), ("max_depth", C.c_int), ("device_bytes", C.c_int64),
        ("wide_nodes", C.c_int64),
    ]


# every symbol include/rto.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "rto_version": (C.c_char_p, []),
    "rto_last_error": (C.c_char_p, []),
    "rto_device_count": (C.c_int, []),
    "rto_options_default": (None, [C.POINTER(COptions)]),
    "rto_options_from_json_file": (C.c_int, [C.c_char_p, C.POINTER(COptions)]),
    "rto_options_from_json": (C.c_int, [C.c_char_p, C.POINTER(COptions)]),
    "rto_tree_load_npz": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    "rto_tree_load_npz_ex": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(_P)]),
    "rto_tree_from_arrays": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_char_p,
                                       C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                       C.POINTER(_P)]),
    "rto_tree_from_arrays_ex": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_char_p,
                                          C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int,
                                          C.POINTER(_P)]),
    "rto_tree_set_ndc": (C.c_int, [_P, C.c_float, C.c_float, C.c_float]),
    "rto_tree_get_info": (C.c_int, [_P, C.POINTER(CTreeInfo)]),
    "rto_tree_probe_npz": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t]),
    "rto_tree_free": (None, [_P]),
    "rto_ctx_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "rto_ctx_create_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "rto_ctx_frames": (C.c_int, [_P]),
    "rto_ctx_select_frame": (C.c_int, [_P, C.c_int]),
    "rto_ctx_free": (None, [_P]),
    "rto_ctx_width": (C.c_int, [_P]),
    "rto_ctx_height": (C.c_int, [_P]),
    "rto_ctx_aux": (_P, [_P]),
    "rto_ctx_noisy": (_P, [_P]),
    "rto_ctx_image": (_P, [_P]),
    "rto_ctx_rng_seed": (None, [_P, C.c_uint64, C.c_uint64]),
    "rto_ctx_rng_advance": (None, [_P, C.c_int64]),
    "rto_ctx_rng_set": (None, [_P, C.c_uint64, C.c_uint64]),
    "rto_ctx_rng_get": (None, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rto_ctx_set_kernel": (C.c_int, [_P, C.c_int]),
    "rto_ctx_set_tuning": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "rto_ctx_set_lean_outputs": (C.c_int, [_P, C.c_int]),
    "rto_ctx_frames_are_lean": (C.c_int, [_P, C.c_int, C.c_int]),
    "rto_ctx_frames_lean_level": (C.c_int, [_P, C.c_int, C.c_int]),
    "rto_ctx_queue_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rto_ctx_tile_marks": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "rto_ctx_kernel_timing": (C.c_int, [_P, C.c_int]),
    "rto_ctx_kernel_timing_read": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "rto_ctx_kernel_timing_read3": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "rto_ctx_enable_stats": (C.c_int, [_P, C.c_int]),
    "rto_ctx_get_stats": (C.c_int, [_P, _P, C.POINTER(C.c_uint64), C.c_int]),
    "rto_ctx_get_march_stats": (C.c_int, [_P, _P, C.POINTER(C.c_uint64), C.c_int]),
    "rto_wide_image_probe": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, _P, C.c_int64, _P, _P, _P, C.POINTER(C.c_int64)]),
    "rto_launch_renderer": (C.c_int, [_P, C.POINTER(CCamera), C.POINTER(COptions), _P, _P]),
    "rto_launch_renderer_batch": (C.c_int, [_P, C.POINTER(CCamera), C.POINTER(C.c_int64), C.c_int, C.POINTER(COptions), _P, _P]),
    "rto_filtering_batch": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rto_filtering_batch_mode": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int]),
    "rto_filtering_train_forward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "rto_filtering_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rto_filtering": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rto_ctx_filtering": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "rto_ctx_download_rgba8": (C.c_int, [_P, _P, C.c_int, _P]),
    "rto_ctx_download_image": (C.c_int, [_P, _P, C.c_int, _P]),
    "rto_ctx_download_aux": (C.c_int, [_P, _P, _P]),
    "rto_guidance_net_create": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "rto_guidance_net_forward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rto_guidance_net_forward_ex": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int]),
    "rto_guidance_net_forward_packed": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rto_filtering_packed": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "rto_guidance_net_forward_culled": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_int, C.c_float]),
    "rto_filtering_culled": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_int, C.c_float]),
    "rto_guidance_net_forward_packed_culled": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_float]),
    "rto_filtering_packed_culled": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_float]),
    "rto_denoise": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "rto_ctx_selected_frame": (C.c_int, [_P]),
    "rto_guidance_net_reserve": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "rto_guidance_net_free": (None, [_P]),
    "rto_probe_gather": (C.c_int, [C.c_uint64, C.c_int]),
    "rto_probe_gather_sweep": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_double)]),
    "rto_probe_scratch": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "rto_probe_valu": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "rto_probe_valu_name": (C.c_char_p, [C.c_int]),
    "rto_probe_thresholds": (C.c_int, [C.c_uint32, C.c_uint32, _P]),
    "rto_probe_math": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, _P]),
    "rto_probe_sigmoid": (C.c_int, [C.c_int, C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "rto_timer_reset": (C.c_int, [_P, _P]),
    "rto_timer_start": (C.c_int, [_P, C.c_int]),
    "rto_timer_stop": (C.c_int, [_P, C.c_int]),
    "rto_timer_record": (C.c_int, [_P, C.c_int]),
    "rto_timer_report": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
}


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 build of librto.so (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=out)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("librto.so was not produced at " + LIB_PATH)
    return LIB_PATH


_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64.so (soname
    libamdhip64.so.7, the name librto.so needs too).  Whichever copy is mapped first serves both, but
    torch cannot find a GPU if it comes second behind /opt/rocm's -- so map torch's copy before
    librto.so.  Without torch installed this is a no-op and /opt/rocm's runtime is used."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    for d in spec.submodule_search_locations:
        cand = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def lib():
    """The loaded library with prototypes set.  Raises if librto.so is absent or lacks a symbol."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "librto.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the render path)" % LIB_PATH)
        _preload_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            if os.environ.get("RTO_LIB") and os.environ.get("RTO_LIB_OLDER_BUILD") and not hasattr(L, name):
                continue  # (same-box A/B against the library of an OLDER revision, tools/ab_rev.sh: symbols added since are absent)
            fn = getattr(L, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != RTO_OK:
        raise RtoError(rc, lib().rto_last_error().decode("utf-8", "replace"))
    return rc
