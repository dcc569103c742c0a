"""ctypes binding of the product C-ABI (include/vs_stab.h, libvideo-stab.so).

This is plumbing for tests/bench: every call goes straight into the HIP
library.  There is no Python or CPU fallback - if the library is missing, or
no GPU is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG_DIR, "csrc", "libvideo-stab.so")

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)

FMT_BGR8, FMT_NV12, FMT_GRAY8 = 0, 1, 2
BORDER_BLACK, BORDER_REFLECT, BORDER_REFLECT_101, BORDER_REPLICATE, BORDER_WRAP, BORDER_FADE = range(6)
SMOOTH_BOX, SMOOTH_GAUSSIAN, SMOOTH_KALMAN = range(3)
STAGE_WARP, STAGE_WARP_TABLES, STAGE_COUNT = 7, 8, 9      # vs_stab.h VS_STAGE_*


class VsParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("logging", C.c_int32), ("smoothing_radius", C.c_int32),
        ("max_corners", C.c_int32), ("quality_level", C.c_double), ("min_distance", C.c_double),
        ("block_size", C.c_int32), ("border_type", C.c_int32), ("border_size", C.c_int32),
        ("crop_n_zoom", C.c_int32), ("smoothing_method", C.c_int32), ("horizon_lock", C.c_int32),
        ("gaussian_sigma", C.c_double), ("adaptive_smoothing", C.c_int32),
        ("min_smoothing_radius", C.c_int32), ("max_smoothing_radius", C.c_int32),
        ("fade_alpha", C.c_float), ("fade_duration", C.c_int32), ("enable_virtual_canvas", C.c_int32),
        ("drone_high_freq_mode", C.c_int32), ("hf_shake_px", C.c_float),
        ("hf_analysis_max_width", C.c_int32), ("hf_rot_lp_alpha", C.c_float),
        ("enable_conditional_clahe", C.c_int32), ("hf_dead_zone_threshold", C.c_float),
        ("hf_freeze_duration", C.c_int32), ("hf_motion_accumulator_decay", C.c_float),
        ("lk_win_size", C.c_int32), ("lk_max_level", C.c_int32), ("lk_max_iters", C.c_int32),
        ("lk_epsilon", C.c_double), ("ransac_max_iters", C.c_int32), ("ransac_threshold", C.c_double),
        ("canvas_scale_factor", C.c_float), ("temporal_buffer_size", C.c_int32), ("canvas_blend_weight", C.c_float),
        ("adaptive_canvas_size", C.c_int32), ("max_canvas_scale", C.c_float), ("min_canvas_scale", C.c_float),
        ("edge_blend_radius", C.c_int32), ("reserved", C.c_int32 * 1),
    ]


class VsCounters(C.Structure):
    _fields_ = [
        ("frames_in", C.c_uint64), ("frames_out", C.c_uint64), ("detections", C.c_uint64),
        ("last_features", C.c_int32), ("last_tracked", C.c_int32), ("last_inliers", C.c_int32),
        ("last_candidates", C.c_int32), ("gftt_overflow", C.c_int32), ("reserved", C.c_int32 * 7),
    ]


class VsDebugFrame(C.Structure):
    _fields_ = [
        ("n_prev", C.c_int32), ("n_valid", C.c_int32), ("ransac_best_iter", C.c_int32),
        ("ransac_iters_run", C.c_int32), ("n_inliers", C.c_int32), ("detected", C.c_int32),
        ("n_detected", C.c_int32), ("box_radius", C.c_int32), ("intent", C.c_int32),
        ("out_index", C.c_int32), ("transform", C.c_float * 3), ("smoothed", C.c_float * 3),
        ("warp_matrix", C.c_float * 6), ("model", C.c_double * 6),
    ]


class VsRollParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("canny_aperture", C.c_int32), ("scale_factor", C.c_double),
        ("canny_threshold_low", C.c_double), ("canny_threshold_high", C.c_double),
        ("hough_rho", C.c_float), ("hough_theta", C.c_float), ("hough_threshold", C.c_int32),
        ("reserved0", C.c_int32), ("angle_filter_min", C.c_double), ("angle_filter_max", C.c_double),
        ("angle_smoothing_alpha", C.c_double), ("angle_decay", C.c_double), ("max_angle_change_deg", C.c_double),
    ]


class VsEnhParams(C.Structure):
    """vs_enh_params_c (include/vs_stab.h): flat mirror of vs::Enhancer::Parameters."""
    _fields_ = [
        ("struct_size", C.c_int32), ("brightness", C.c_float), ("contrast", C.c_float),
        ("enable_white_balance", C.c_int32), ("wb_strength", C.c_float),
        ("enable_vibrance", C.c_int32), ("vibrance_strength", C.c_float),
        ("enable_unsharp", C.c_int32), ("sharpness", C.c_float), ("blur_sigma", C.c_float),
        ("enable_clahe", C.c_int32), ("clahe_clip_limit", C.c_float), ("clahe_tile_grid_size", C.c_int32),
        ("enable_denoise", C.c_int32), ("denoise_strength", C.c_float), ("gamma", C.c_float),
        ("use_cuda", C.c_int32), ("reserved0", C.c_int32),
    ]


class VsError(RuntimeError):
    pass


def _p(a, t):
    return a.ctypes.data_as(t)


class DevBuf:
    """A device allocation owned through vs_dev_malloc/vs_dev_free."""

    def __init__(self, vs, nbytes):
        self.vs = vs
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        vs.check(vs.lib.vs_dev_malloc(C.byref(p), max(self.nbytes, 16)))
        self.ptr = p.value

    @classmethod
    def from_array(cls, vs, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(vs, arr.nbytes)
        b.upload(arr)
        return b

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= max(self.nbytes, 16)
        self.vs.check(self.vs.lib.vs_dev_memcpy_h2d(C.c_void_p(self.ptr + offset), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def download(self, shape, dtype, offset=0):
        out = np.empty(shape, dtype)
        assert offset + out.nbytes <= max(self.nbytes, 16)
        self.vs.check(self.vs.lib.vs_dev_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + offset), out.nbytes))
        return out

    def zero(self):
        self.vs.check(self.vs.lib.vs_dev_memset(C.c_void_p(self.ptr), 0, max(self.nbytes, 16)))

    def free(self):
        if self.ptr:
            self.vs.lib.vs_dev_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostBuf:
    """Page-locked host memory (vs_host_alloc) as a numpy array: frames that the host entry points move by DMA."""

    def __init__(self, vs, shape, dtype=np.uint8):
        self.vs = vs
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        vs.check(vs.lib.vs_host_alloc(C.byref(p), max(n, 1)))
        self.ptr = p.value
        self.array = np.frombuffer((C.c_uint8 * max(n, 1)).from_address(self.ptr), dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            self.vs.lib.vs_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class VsLib:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        vp = C.c_void_p
        L.vs_abi_version.restype = C.c_int
        try:
            L.vs_build_tag.restype = C.c_char_p
        except AttributeError:      # a library of an earlier build (A/B measurements)
            pass
        L.vs_build_info.restype = C.c_char_p
        L.vs_device_count.restype = C.c_int
        L.vs_params_default.argtypes = [C.POINTER(VsParams)]
        L.vs_status_string.restype = C.c_char_p
        L.vs_status_string.argtypes = [C.c_int]
        L.vs_last_error.restype = C.c_char_p
        L.vs_stab_create.argtypes = [C.POINTER(VsParams), C.c_int, C.POINTER(vp)]
        L.vs_stab_destroy.argtypes = [vp]
        L.vs_stab_clean.argtypes = [vp]
        L.vs_stab_push.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, u8p, C.c_size_t, i32p]
        L.vs_stab_flush.argtypes = [vp, u8p, C.c_size_t, i32p]
        L.vs_stab_push_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_int, vp, C.c_size_t, i32p]
        L.vs_stab_flush_dev.argtypes = [vp, vp, C.c_size_t, i32p]
        L.vs_stab_sync.argtypes = [vp]
        L.vs_stab_out_size.argtypes = [vp, C.c_int, C.c_int, i32p, i32p]
        L.vs_stab_last_out_dims.argtypes = [vp, i32p, i32p]
        L.vs_stab_get_counters.argtypes = [vp, C.POINTER(VsCounters)]
        L.vs_stab_get_debug.argtypes = [vp, C.POINTER(VsDebugFrame)]
        L.vs_stab_canvas_info.argtypes = [vp, i32p]
        L.vs_stab_get_debug_arrays.argtypes = [vp, f32p, f32p, u8p, u8p, f32p, u8p, i32p, i32p]
        L.vs_stab_last_error.restype = C.c_char_p
        L.vs_stab_last_error.argtypes = [vp]
        L.vs_stab_stream.restype = vp
        L.vs_stab_stream.argtypes = [vp]
        L.vs_stab_set_warp_batch.argtypes = [vp, C.c_int]
        L.vs_stab_set_batch.argtypes = [vp, C.c_int]
        L.vs_stab_set_zero_copy.argtypes = [vp, C.c_int]
        L.vs_stab_set_nv12_layout.argtypes = [vp, C.c_size_t, C.c_size_t]
        L.vs_stab_set_profiling.argtypes = [vp, C.c_int]
        L.vs_stab_get_stage_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        try:
            L.vs_batch_create.argtypes = [C.c_int, C.c_int, C.POINTER(VsParams), C.c_int, C.POINTER(vp)]
            L.vs_batch_create_params.argtypes = [C.c_int, C.c_int, C.POINTER(VsParams), C.c_int, C.POINTER(vp)]
            L.vs_batch_destroy.argtypes = [vp]
            L.vs_batch_destroy.restype = None
            L.vs_batch_streams.argtypes = [vp]
            L.vs_batch_stream.argtypes = [vp, C.c_int]
            L.vs_batch_stream.restype = vp
            L.vs_batch_set_zero_copy.argtypes = [vp, C.c_int]
            L.vs_batch_set_nv12_layout.argtypes = [vp, C.c_size_t, C.c_size_t]
            L.vs_batch_push_dev.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_size_t, C.c_int, C.POINTER(vp), C.c_size_t, i32p]
            L.vs_batch_flush_dev.argtypes = [vp, C.POINTER(vp), C.c_size_t, i32p]
            L.vs_batch_sync.argtypes = [vp]
            L.vs_batch_last_error.argtypes = [vp]
            L.vs_batch_last_error.restype = C.c_char_p
        except AttributeError:      # a library of an earlier build (A/B measurements)
            pass
        L.vs_dev_set_device.argtypes = [C.c_int]
        L.vs_dev_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
        try:
            L.vs_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
            L.vs_host_free.argtypes = [vp]
            L.vs_host_free.restype = None
            L.vs_stab_set_host_pipeline.argtypes = [vp, C.c_int]
        except AttributeError:      # a library of an earlier build (A/B measurements)
            pass
        L.vs_dev_free.argtypes = [vp]
        L.vs_dev_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
        L.vs_dev_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
        L.vs_dev_memset.argtypes = [vp, C.c_int, C.c_size_t]
        L.vs_dev_memcpy_d2d.argtypes = [vp, vp, C.c_size_t]
        L.vs_dev_copy_rate.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        L.vs_op_libm_checksum.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        L.vs_op_warp_affine.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                        C.c_int, f32p, C.c_int, vp]
        L.vs_op_warp_affine_nv12.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int, f32p, C.c_int,
                                             C.c_size_t, C.c_size_t, vp]
        L.vs_op_resize_gray.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, C.c_int, C.c_int, vp]
        L.vs_op_pyr_down.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, C.c_size_t, vp]
        L.vs_op_scharr.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
        L.vs_op_pyr_level.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int, vp, vp, C.c_size_t, C.c_size_t, C.c_int, vp]
        L.vs_op_pyr_lk.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp,
                                   C.c_int, C.c_int, C.c_int, C.c_double, vp]
        L.vs_op_gftt.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                 vp, vp, vp, vp]
        L.vs_op_estimate_affine_partial2d.argtypes = [vp, vp, C.c_int, C.c_double, C.c_int, vp, vp, vp, vp]
        L.vs_roll_params_default.argtypes = [C.POINTER(VsRollParams)]
        L.vs_roll_params_default.restype = None
        L.vs_roll_create.argtypes = [C.POINTER(VsRollParams), C.c_int, C.POINTER(vp)]
        L.vs_roll_destroy.argtypes = [vp]
        L.vs_roll_destroy.restype = None
        L.vs_roll_last_error.argtypes = [vp]
        L.vs_roll_last_error.restype = C.c_char_p
        L.vs_roll_correct.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t]
        L.vs_roll_correct_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t]
        L.vs_roll_correct_nv12_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t]
        L.vs_roll_correct_nv12_dev_n.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
        L.vs_azc_apply_nv12_dev_n.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                              C.POINTER(C.c_int64)]
        L.vs_stab_push_dev_n.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.POINTER(vp), C.c_size_t, i32p]
        L.vs_azc_apply_nv12_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_int64)]
        L.vs_azc_result.argtypes = [vp, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int), i32p]
        L.vs_roll_sync.argtypes = [vp]
        L.vs_roll_get_state.argtypes = [vp, f64p, f64p, i32p, i32p]
        L.vs_op_canny.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double, vp, C.c_size_t, vp]
        L.vs_op_hough_lines.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, vp,
                                        C.c_int, vp, vp]
        L.vs_azc_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.vs_azc_destroy.argtypes = [vp]
        L.vs_azc_destroy.restype = None
        L.vs_azc_last_error.argtypes = [vp]
        L.vs_azc_last_error.restype = C.c_char_p
        L.vs_azc_apply.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, u8p, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]
        L.vs_azc_apply_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_int, vp, C.c_size_t,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vs_azc_sync.argtypes = [vp]
        L.vs_azc_worker_times.argtypes = [vp, C.POINTER(C.c_double)]
        L.vs_azc_get_info.argtypes = [vp, i32p]
        L.vs_op_content_mask.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]
        L.vs_azc_crop_from_mask.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, i32p, u8p]
        L.vs_op_external_boxes.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, i32p, C.c_int, i32p]
        cp = C.c_char_p
        L.vs_config_open.argtypes = [cp, C.POINTER(vp)]
        L.vs_config_parse.argtypes = [cp, C.c_size_t, C.POINTER(vp)]
        L.vs_config_close.argtypes = [vp]
        L.vs_config_close.restype = None
        L.vs_config_kind.argtypes = [vp, cp]
        L.vs_config_size.argtypes = [vp, cp]
        L.vs_config_get_int.argtypes = [vp, cp, i32p]
        L.vs_config_get_double.argtypes = [vp, cp, f64p]
        L.vs_config_get_float.argtypes = [vp, cp, C.POINTER(C.c_float)]
        L.vs_config_get_string.argtypes = [vp, cp, C.c_char_p, C.c_size_t]
        L.vs_config_seq_get_double.argtypes = [vp, cp, C.c_int, f64p]
        L.vs_config_read_stab.argtypes = [vp, cp, C.c_int, C.POINTER(VsParams), i32p]
        L.vs_config_read_roll.argtypes = [vp, cp, C.c_int, C.POINTER(VsRollParams), i32p]
        L.vs_config_read_enh.argtypes = [vp, cp, C.c_int, C.POINTER(VsEnhParams), i32p]
        L.vs_config_mtime.argtypes = [cp, C.POINTER(C.c_int64)]
        ep = C.POINTER(VsEnhParams)
        L.vs_enh_params_default.argtypes = [ep]
        L.vs_enh_params_default.restype = None
        L.vs_enh_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.vs_enh_destroy.argtypes = [vp]
        L.vs_enh_destroy.restype = None
        L.vs_enh_last_error.argtypes = [vp]
        L.vs_enh_last_error.restype = C.c_char_p
        L.vs_enh_apply.argtypes = [vp, ep, u8p, C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t]
        L.vs_enh_apply_dev.argtypes = [vp, ep, vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t]
        L.vs_enh_apply_batch_dev.argtypes = [vp, ep, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.c_int,
                                             C.c_size_t, C.c_size_t]
        L.vs_enh_sync.argtypes = [vp]
        L.vs_enh_last_passes.argtypes = [vp]
        L.vs_enh_cvt_color.argtypes = [vp, C.c_int, vp, vp, C.c_size_t]
        L.vs_enh_gaussian_blur.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_double, vp, C.c_size_t]
        L.vs_op_warp_affine_ex.argtypes =[vp, C.c_size_t, C.c_int, C.c_int, vp, C.c_size_t, C.c_int, C.c_int,
                                           C.c_int, f64p, C.c_int, vp]

    # ---- helpers ----------------------------------------------------------
    def check(self, status, inst=None):
        if status != 0:
            msg = self.lib.vs_status_string(status).decode()
            detail = (self.lib.vs_stab_last_error(inst) if inst else self.lib.vs_last_error()) or b""
            raise VsError("%s: %s" % (msg, detail.decode()))

    def params(self, **kw):
        p = VsParams()
        self.lib.vs_params_default(C.byref(p))
        for k, v in kw.items():
            assert hasattr(p, k), k
            setattr(p, k, v)
        return p

    def sync(self):
        self.check(self.lib.vs_dev_sync())

    # ---- stage operators, numpy in / numpy out (device round trip) ----------
    def warp_affine(self, img, M, batch=None):
        """img: (h,w[,3]) or (b,h,w,3) uint8; M: (6,) or (b,6).  batch=True / False settles what a 3-d array is: b planes of
        one channel, or one image of shape[2] channels."""
        img = np.ascontiguousarray(img)
        batched = img.ndim == 4 or (img.ndim == 3 and (img.shape[2] not in (1, 3) if batch is None else batch))
        if img.ndim == 2:
            b, (h, w), cn = 1, img.shape, 1
        elif img.ndim == 3 and not batched:
            b, (h, w, cn) = 1, img.shape
        elif img.ndim == 3:
            (b, h, w), cn = img.shape, 1
        else:
            b, h, w, cn = img.shape
        M = np.ascontiguousarray(M, np.float32).reshape(b, 6)
        d_in = DevBuf.from_array(self, img)
        d_out = DevBuf(self, img.nbytes)
        fb = h * w * cn
        self.check(self.lib.vs_op_warp_affine(d_in.ptr, w * cn, fb, d_out.ptr, w * cn, fb, w, h, cn,
                                              _p(M, f32p), b, None))
        self.sync()
        return d_out.download(img.shape, np.uint8)

    def warp_affine_nv12(self, img, w, h, M):
        """img: one NV12 surface (h * 3 / 2 rows of w bytes) and one matrix, or a stack of n surfaces and n matrices (launches
        of four and more surfaces warp both planes in one grid)."""
        img = np.ascontiguousarray(img)
        M = np.ascontiguousarray(M, np.float32).reshape(-1, 6)
        n = M.shape[0]
        fb = img.nbytes // n
        d_in = DevBuf.from_array(self, img)
        d_out = DevBuf(self, img.nbytes)
        self.check(self.lib.vs_op_warp_affine_nv12(d_in.ptr, w, d_out.ptr, w, w, h, _p(M, f32p), n, fb, fb, None))
        self.sync()
        return d_out.download(img.shape, np.uint8)

    def resize_gray(self, img, dw, dh, fmt=None):
        img = np.ascontiguousarray(img)
        if fmt is None:
            fmt = FMT_BGR8 if img.ndim == 3 else FMT_GRAY8
        w = img.shape[1]
        h = img.shape[0] if fmt != FMT_NV12 else img.shape[0] * 2 // 3
        cn = 3 if fmt == FMT_BGR8 else 1
        d_in = DevBuf.from_array(self, img)
        d_out = DevBuf(self, dw * dh)
        self.check(self.lib.vs_op_resize_gray(d_in.ptr, w * cn, w, h, fmt, d_out.ptr, dw, dw, dh, None))
        self.sync()
        return d_out.download((dh, dw), np.uint8)

    def pyr_down(self, g):
        g = np.ascontiguousarray(g)
        h, w = g.shape
        dh, dw = (h + 1) // 2, (w + 1) // 2
        d_in = DevBuf.from_array(self, g)
        d_out = DevBuf(self, dw * dh)
        self.check(self.lib.vs_op_pyr_down(d_in.ptr, w, w, h, d_out.ptr, dw, None))
        self.sync()
        return d_out.download((dh, dw), np.uint8)

    def scharr(self, g):
        g = np.ascontiguousarray(g)
        h, w = g.shape
        d_in = DevBuf.from_array(self, g)
        d_out = DevBuf(self, h * w * 4)
        self.check(self.lib.vs_op_scharr(d_in.ptr, w, w, h, d_out.ptr, None))
        self.sync()
        return d_out.download((h, w, 2), np.int16)

    def pyr_level(self, g, down=True):
        """One pyramid level of batch mode: (derivatives int16 (h, w, 2), next level or None) of a gray image."""
        g = np.ascontiguousarray(g)
        h, w = g.shape
        dh, dw = (h + 1) // 2, (w + 1) // 2
        d_in = DevBuf.from_array(self, g)
        d_der = DevBuf(self, h * w * 4)
        d_next = DevBuf(self, dw * dh) if down else None
        self.check(self.lib.vs_op_pyr_level(d_in.ptr, w, w * h, w, h, d_der.ptr, d_next.ptr if down else None, dw, dw * dh, 1, None))
        self.sync()
        return d_der.download((h, w, 2), np.int16), (d_next.download((dh, dw), np.uint8) if down else None)

    def pyr_lk(self, prev, nxt, pts, win=15, max_level=2, iters=20, eps=0.03):
        prev = np.ascontiguousarray(prev)
        nxt = np.ascontiguousarray(nxt)
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        n = pts.shape[0]
        h, w = prev.shape
        d_prev = DevBuf.from_array(self, prev)
        d_next = DevBuf.from_array(self, nxt)
        d_pts = DevBuf.from_array(self, pts)
        d_out = DevBuf(self, n * 8)
        d_st = DevBuf(self, n)
        d_err = DevBuf(self, n * 4)
        self.check(self.lib.vs_op_pyr_lk(d_prev.ptr, d_next.ptr, w, w, h, d_pts.ptr, n, d_out.ptr, d_st.ptr,
                                         d_err.ptr, win, max_level, iters, eps, None))
        self.sync()
        return (d_out.download((n, 2), np.float32), d_st.download((n,), np.uint8),
                d_err.download((n,), np.float32))

    def gftt(self, g, max_corners, quality, min_distance, block_size=3, want_eig=False):
        g = np.ascontiguousarray(g)
        h, w = g.shape
        d_in = DevBuf.from_array(self, g)
        d_pts = DevBuf(self, max(max_corners, 1) * 8)
        d_cnt = DevBuf(self, 16)
        d_eig = DevBuf(self, w * h * 4) if want_eig else None
        self.check(self.lib.vs_op_gftt(d_in.ptr, w, w, h, max_corners, quality, min_distance, block_size,
                                       d_pts.ptr, d_cnt.ptr, d_eig.ptr if d_eig else None, None))
        self.sync()
        n = int(d_cnt.download((1,), np.int32)[0])
        pts = d_pts.download((max(max_corners, 1), 2), np.float32)[:n].copy()
        if want_eig:
            return pts, d_eig.download((h, w), np.float32)
        return pts

    def estimate_affine_partial2d(self, a, b, thr=5.0, max_iters=500):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(b, np.float32).reshape(-1, 2)
        n = a.shape[0]
        d_a = DevBuf.from_array(self, a)
        d_b = DevBuf.from_array(self, b)
        d_model = DevBuf(self, 48)
        d_inl = DevBuf(self, max(n, 1))
        d_info = DevBuf(self, 16)
        self.check(self.lib.vs_op_estimate_affine_partial2d(d_a.ptr, d_b.ptr, n, thr, max_iters, d_model.ptr,
                                                            d_inl.ptr, d_info.ptr, None))
        self.sync()
        info = d_info.download((4,), np.int32)
        return (int(info[0]), d_model.download((6,), np.float64), d_inl.download((max(n, 1),), np.uint8)[:n], info)

    def canny(self, g, low, high):
        g = np.ascontiguousarray(g)
        h, w = g.shape
        d_in = DevBuf.from_array(self, g)
        d_out = DevBuf(self, w * h)
        self.check(self.lib.vs_op_canny(d_in.ptr, w, w, h, low, high, d_out.ptr, w, None))
        self.sync()
        return d_out.download((h, w), np.uint8)

    def hough_lines(self, edges, rho, theta, threshold, max_lines=8192):
        edges = np.ascontiguousarray(edges)
        h, w = edges.shape
        d_in = DevBuf.from_array(self, edges)
        d_lines = DevBuf(self, max_lines * 8)
        d_cnt = DevBuf(self, 16)
        self.check(self.lib.vs_op_hough_lines(d_in.ptr, w, w, h, rho, theta, threshold, d_lines.ptr, max_lines,
                                              d_cnt.ptr, None))
        self.sync()
        n = int(d_cnt.download((1,), np.int32)[0])
        return d_lines.download((max_lines, 2), np.float32)[:n].copy()

    def warp_affine_ex(self, img, M, border=BORDER_BLACK, dsize=None):
        """cv::warpAffine with a double 2x3 forward matrix and a border mode."""
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        dw, dh = dsize if dsize else (w, h)
        M = np.ascontiguousarray(M, np.float64).reshape(6)
        d_in = DevBuf.from_array(self, img)
        d_out = DevBuf(self, dw * dh * cn)
        self.check(self.lib.vs_op_warp_affine_ex(d_in.ptr, w * cn, w, h, d_out.ptr, dw * cn, dw, dh, cn,
                                                 _p(M, f64p), border, None))
        self.sync()
        return d_out.download((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)

    def content_mask(self, img):
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        cn = 1 if img.ndim == 2 else 3
        d_in = DevBuf.from_array(self, img)
        d_out = DevBuf(self, w * h)
        self.check(self.lib.vs_op_content_mask(d_in.ptr, w * cn, w, h, cn, d_out.ptr, w, None))
        self.sync()
        return d_out.download((h, w), np.uint8)

    def azc_crop_from_mask(self, mask, want_filled=False):
        """Host logic of AutoZoomCrop (no device needed): info8 [, filled mask]."""
        mask = np.ascontiguousarray(mask)
        h, w = mask.shape
        info = np.zeros(8, np.int32)
        filled = np.empty((h, w), np.uint8) if want_filled else None
        self.check(self.lib.vs_azc_crop_from_mask(_p(mask, u8p), w, h, w, _p(info, i32p),
                                                  _p(filled, u8p) if want_filled else None))
        return (info, filled) if want_filled else info

    def external_boxes(self, mask):
        """Bounding rectangles (x, y, w, h) of the external contours of a host mask, in cv::findContours order."""
        mask = np.ascontiguousarray(mask)
        h, w = mask.shape
        cap = w * h // 2 + 4
        xywh = np.zeros((cap, 4), np.int32)
        n = C.c_int32()
        self.check(self.lib.vs_op_external_boxes(_p(mask, u8p), w, h, w, _p(xywh, i32p), cap, C.byref(n)))
        return xywh[:n.value].copy()

    def auto_zoom_crop(self, device=0):
        return AutoZoomCrop(self, device)

    def roll_params(self, **kw):
        p = VsRollParams()
        self.lib.vs_roll_params_default(C.byref(p))
        for k, v in kw.items():
            assert hasattr(p, k), k
            setattr(p, k, v)
        return p

    def roll_correction(self, params=None, device=0):
        return RollCorrection(self, params or self.roll_params(), device)

    def batch(self, params, n_streams, frames_per_step, device=0):
        return Batch(self, params, n_streams, frames_per_step, device)

    def stabilizer(self, params, device=0):
        return Stabilizer(self, params, device)


class AutoZoomCrop:
    """C-ABI mirror of vs::AutoZoomCrop::autoZoomCrop (AutoZoomCrop.cpp:102-283)."""

    def __init__(self, vs, device=0):
        self.vs = vs
        self.lib = vs.lib
        h = C.c_void_p()
        vs.check(self.lib.vs_azc_create(device, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.lib.vs_azc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status):
        if status != 0:
            raise VsError("%s: %s" % (self.lib.vs_status_string(status).decode(),
                                      (self.lib.vs_azc_last_error(self.h) or b"").decode()))

    def apply(self, frame):
        frame = np.ascontiguousarray(frame)
        h, w = frame.shape[:2]
        cn = 1 if frame.ndim == 2 else 3
        buf = np.empty(max(w * h, 640 * 360) * cn, np.uint8)
        ow, oh = C.c_int(), C.c_int()
        self._check(self.lib.vs_azc_apply(self.h, _p(frame, u8p), w, h, w * cn, cn, _p(buf, u8p),
                                          C.byref(ow), C.byref(oh)))
        shape = (oh.value, ow.value) if cn == 1 else (oh.value, ow.value, 3)
        return buf[:oh.value * ow.value * cn].reshape(shape).copy()

    def apply_nv12_dev(self, d_in, w, h, pitch, d_out, out_pitch, out_uv_offset, uv_offset=0):
        """Asynchronous (vs_azc_apply_nv12_dev): returns the ticket; result(ticket) tells what came out, sync() completes the pixels."""
        t = C.c_int64(-1)
        self._check(self.lib.vs_azc_apply_nv12_dev(self.h, d_in, w, h, pitch, uv_offset, d_out, out_pitch, out_uv_offset, C.byref(t)))
        return t.value

    def apply_nv12_dev_n(self, d_ins, w, h, pitch, d_outs, out_pitch, out_uv_offset, uv_offset=0):
        """n surfaces in call order, one trip through the binding; returns the tickets."""
        n = len(d_ins)
        a = (C.c_void_p * n)(*d_ins)
        b = (C.c_void_p * n)(*d_outs)
        t = (C.c_int64 * n)()
        self._check(self.lib.vs_azc_apply_nv12_dev_n(self.h, a, b, n, w, h, pitch, uv_offset, out_pitch, out_uv_offset, t))
        return list(t)

    def result(self, ticket):
        ow, oh = C.c_int(), C.c_int()
        info = np.zeros(8, np.int32)
        self._check(self.lib.vs_azc_result(self.h, ticket, C.byref(ow), C.byref(oh), _p(info, i32p)))
        return ow.value, oh.value, info

    def apply_dev(self, d_in, w, h, stride, cn, d_out, out_stride):
        ow, oh = C.c_int(), C.c_int()
        self._check(self.lib.vs_azc_apply_dev(self.h, d_in, w, h, stride, cn, d_out, out_stride,
                                              C.byref(ow), C.byref(oh)))
        return ow.value, oh.value

    def sync(self):
        self._check(self.lib.vs_azc_sync(self.h))

    def worker_times(self):
        """(frames, s without a frame, s waiting for masks, s in the contour logic, s queueing / publishing - summed over the workers;
        batches, s from launches to masks, s from masks to the crop launch - summed over the batches; s the caller waited for a slot)"""
        out = (C.c_double * 9)()
        self._check(self.lib.vs_azc_worker_times(self.h, out))
        return tuple(out)

    def info(self):
        info = np.zeros(8, np.int32)
        self._check(self.lib.vs_azc_get_info(self.h, _p(info, i32p)))
        return info


class Config:
    """The config layer of the C ABI (vs_config_*): config.yaml of the reference's example mains, read with the
    conversions of cv::FileNode (examples/vs.cpp:50-168, examples/vsg.cpp:1007-1112).  Host only."""

    KINDS = ("none", "int", "real", "string", "map", "seq")

    def __init__(self, vs, path=None, text=None):
        self.vs = vs
        self.lib = vs.lib
        h = C.c_void_p()
        if path is not None:
            vs.check(self.lib.vs_config_open(os.fsencode(path), C.byref(h)))
        else:
            raw = text.encode() if isinstance(text, str) else bytes(text)
            vs.check(self.lib.vs_config_parse(raw, len(raw), C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.lib.vs_config_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def kind(self, key):
        return self.KINDS[self.lib.vs_config_kind(self.h, key.encode())]

    def size(self, key):
        return self.lib.vs_config_size(self.h, key.encode())

    def get_int(self, key):
        v = C.c_int32()
        self.vs.check(self.lib.vs_config_get_int(self.h, key.encode(), C.byref(v)))
        return v.value

    def get_bool(self, key):
        return self.get_int(key) != 0

    def get_double(self, key):
        v = C.c_double()
        self.vs.check(self.lib.vs_config_get_double(self.h, key.encode(), C.byref(v)))
        return v.value

    def get_float(self, key):
        v = C.c_float()
        self.vs.check(self.lib.vs_config_get_float(self.h, key.encode(), C.byref(v)))
        return v.value

    def get_string(self, key):
        buf = C.create_string_buffer(4096)
        self.vs.check(self.lib.vs_config_get_string(self.h, key.encode(), buf, len(buf)))
        return buf.value.decode()

    def get_seq(self, key):
        out = []
        for i in range(self.size(key) if self.kind(key) == "seq" else 0):
            v = C.c_double()
            self.vs.check(self.lib.vs_config_seq_get_double(self.h, key.encode(), i, C.byref(v)))
            out.append(v.value)
        return out

    def _read(self, fn, p, section, zero_missing):
        present = C.c_int32()
        self.vs.check(fn(self.h, section.encode(), int(zero_missing), C.byref(p), C.byref(present)))
        return p, bool(present.value)

    def stab_params(self, section="stabilizer", zero_missing=False, base=None):
        return self._read(self.lib.vs_config_read_stab, base or self.vs.params(), section, zero_missing)

    def roll_params(self, section="roll_correction", zero_missing=False, base=None):
        return self._read(self.lib.vs_config_read_roll, base or self.vs.roll_params(), section, zero_missing)

    def enh_params(self, section="enhancer", zero_missing=False, base=None):
        return self._read(self.lib.vs_config_read_enh, base or Enhancer.default_params(self.vs), section, zero_missing)

    @staticmethod
    def mtime(vs, path):
        v = C.c_int64()
        vs.check(vs.lib.vs_config_mtime(os.fsencode(path), C.byref(v)))
        return v.value


class Enhancer:
    """C-ABI mirror of vs::Enhancer::enhanceImage (Enhancer.cpp:138-239)."""

    CVT = dict(bgr2hsv=0, hsv2bgr=1, bgr2lab=2, lab2bgr=3)

    def __init__(self, vs, device=0):
        self.vs = vs
        self.lib = vs.lib
        h = C.c_void_p()
        vs.check(self.lib.vs_enh_create(device, C.byref(h)))
        self.h = h

    @staticmethod
    def default_params(vs, **kw):
        p = VsEnhParams()
        vs.lib.vs_enh_params_default(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        return p

    def close(self):
        if self.h:
            self.lib.vs_enh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status):
        if status != 0:
            raise VsError("%s: %s" % (self.lib.vs_status_string(status).decode(),
                                      (self.lib.vs_enh_last_error(self.h) or b"").decode()))

    def apply(self, frame, params):
        frame = np.ascontiguousarray(frame)
        h, w = frame.shape[:2]
        out = np.empty_like(frame)
        self._check(self.lib.vs_enh_apply(self.h, C.byref(params), _p(frame, u8p), w, h, w * 3, _p(out, u8p), w * 3))
        return out

    def apply_dev(self, params, d_in, w, h, stride, d_out, out_stride):
        self._check(self.lib.vs_enh_apply_dev(self.h, C.byref(params), d_in, w, h, stride, d_out, out_stride))

    def apply_batch_dev(self, params, d_ins, d_outs, w, h, stride, out_stride):
        n = len(d_ins)
        a = (C.c_void_p * n)(*[C.c_void_p(int(x)) for x in d_ins])
        b = (C.c_void_p * n)(*[C.c_void_p(int(x)) for x in d_outs])
        self._check(self.lib.vs_enh_apply_batch_dev(self.h, C.byref(params), a, b, n, w, h, stride, out_stride))

    def sync(self):
        self._check(self.lib.vs_enh_sync(self.h))

    def passes(self):
        return self.lib.vs_enh_last_passes(self.h)

    def cvt_color(self, code, px):
        """px: (n,3) uint8 host array -> converted (n,3) array (through device buffers)."""
        px = np.ascontiguousarray(px, np.uint8)
        n = px.shape[0]
        d_in, d_out = DevBuf(self.vs, n * 3), DevBuf(self.vs, n * 3)
        d_in.upload(px)
        self._check(self.lib.vs_enh_cvt_color(self.h, self.CVT[code], d_in.ptr, d_out.ptr, n))
        self.sync()
        return d_out.download((n, 3), np.uint8)

    def gaussian_blur(self, frame, sigma):
        frame = np.ascontiguousarray(frame)
        h, w = frame.shape[:2]
        d_in, d_out = DevBuf(self.vs, frame.nbytes), DevBuf(self.vs, frame.nbytes)
        d_in.upload(frame)
        self._check(self.lib.vs_enh_gaussian_blur(self.h, d_in.ptr, w * 3, w, h, sigma, d_out.ptr, w * 3))
        self.sync()
        return d_out.download(frame.shape, np.uint8)


class RollCorrection:
    """C-ABI mirror of vs::RollCorrection::autoCorrectRoll (RollCorrection.cpp:16-155)."""

    def __init__(self, vs, params, device=0):
        self.vs = vs
        self.lib = vs.lib
        h = C.c_void_p()
        vs.check(self.lib.vs_roll_create(C.byref(params), device, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.lib.vs_roll_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status):
        if status != 0:
            raise VsError("%s: %s" % (self.lib.vs_status_string(status).decode(),
                                      (self.lib.vs_roll_last_error(self.h) or b"").decode()))

    def correct(self, frame):
        frame = np.ascontiguousarray(frame)
        h, w = frame.shape[:2]
        out = np.empty_like(frame)
        self._check(self.lib.vs_roll_correct(self.h, _p(frame, u8p), w, h, w * 3, _p(out, u8p), w * 3))
        return out

    def correct_dev(self, d_in, w, h, stride, d_out, out_stride):
        self._check(self.lib.vs_roll_correct_dev(self.h, d_in, w, h, stride, d_out, out_stride))

    def correct_nv12_dev_n(self, d_ins, w, h, pitch, d_outs, out_pitch, uv_offset=0, out_uv_offset=0):
        """n surfaces in call order (lists of device pointers), one trip through the binding."""
        n = len(d_ins)
        a = (C.c_void_p * n)(*d_ins)
        b = (C.c_void_p * n)(*d_outs)
        self._check(self.lib.vs_roll_correct_nv12_dev_n(self.h, a, b, n, w, h, pitch, uv_offset, out_pitch, out_uv_offset))

    def correct_nv12_dev(self, d_in, w, h, pitch, d_out, out_pitch, uv_offset=0, out_uv_offset=0):
        """Asynchronous (vs_roll_correct_nv12_dev): the result is complete after sync()."""
        self._check(self.lib.vs_roll_correct_nv12_dev(self.h, d_in, w, h, pitch, uv_offset, d_out, out_pitch, out_uv_offset))

    def sync(self):
        self._check(self.lib.vs_roll_sync(self.h))

    def state(self):
        s, d = C.c_double(), C.c_double()
        n, u = C.c_int32(), C.c_int32()
        self._check(self.lib.vs_roll_get_state(self.h, C.byref(s), C.byref(d), C.byref(n), C.byref(u)))
        return s.value, d.value, n.value, u.value


class Stabilizer:
    """One video stream: the C-ABI mirror of vs::Stabilizer (Stabilizer.h:177-198)."""

    def __init__(self, vs, params, device=0):
        self.vs = vs
        self.lib = vs.lib
        h = C.c_void_p()
        vs.check(self.lib.vs_stab_create(C.byref(params), device, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.lib.vs_stab_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _geom(self, frame, fmt):
        w = frame.shape[1]
        h = frame.shape[0] if fmt != FMT_NV12 else frame.shape[0] * 2 // 3
        return w, h, (3 if fmt == FMT_BGR8 else 1)

    def out_shape(self, w, h, fmt):
        ow, oh = C.c_int32(), C.c_int32()
        self.vs.check(self.lib.vs_stab_out_size(self.h, w, h, C.byref(ow), C.byref(oh)), self.h)
        if fmt == FMT_BGR8:
            return (oh.value, ow.value, 3)
        if fmt == FMT_NV12:
            return (oh.value * 3 // 2, ow.value)
        return (oh.value, ow.value)

    def push(self, frame, fmt=FMT_BGR8, out=None):
        """stabilize(frame): returns the stabilized frame or None (warm-up).  `out`: a caller's array for the result (for
        instance page-locked: HostBuf), else a new one."""
        frame = np.ascontiguousarray(frame)
        w, h, cn = self._geom(frame, fmt)
        if out is None:
            out = np.zeros(self.out_shape(w, h, fmt), np.uint8)
        produced = C.c_int32(0)
        self.vs.check(self.lib.vs_stab_push(self.h, _p(frame, u8p), w, h, w * cn, fmt, _p(out, u8p),
                                            out.shape[1] * cn, C.byref(produced)), self.h)
        return out if produced.value else None

    def flush(self, like, fmt=FMT_BGR8):
        w, h, cn = self._geom(like, fmt)
        out = np.zeros(self.out_shape(w, h, fmt), np.uint8)
        produced = C.c_int32(0)
        self.vs.check(self.lib.vs_stab_flush(self.h, _p(out, u8p), out.shape[1] * cn, C.byref(produced)), self.h)
        return out if produced.value else None

    def set_host_pipeline(self, on=True):
        self.vs.check(self.lib.vs_stab_set_host_pipeline(self.h, int(on)), self.h)

    def push_dev(self, d_in, w, h, stride, fmt, d_out, out_stride):
        produced = C.c_int32(0)
        self.vs.check(self.lib.vs_stab_push_dev(self.h, d_in, w, h, stride, fmt, d_out, out_stride,
                                                C.byref(produced)), self.h)
        return produced.value

    def push_dev_n(self, d_ins, w, h, stride, fmt, d_outs, out_stride):
        """len(d_ins) consecutive pushes in one call; the j-th result that becomes due goes to d_outs[j].  Returns how many did."""
        n = len(d_ins)
        a = (C.c_void_p * n)(*d_ins)
        b = (C.c_void_p * n)(*d_outs)
        produced = C.c_int32(0)
        self.vs.check(self.lib.vs_stab_push_dev_n(self.h, a, n, w, h, stride, fmt, b, out_stride, C.byref(produced)), self.h)
        return produced.value

    def flush_dev(self, d_out, out_stride):
        produced = C.c_int32(0)
        self.vs.check(self.lib.vs_stab_flush_dev(self.h, d_out, out_stride, C.byref(produced)), self.h)
        return produced.value

    def sync(self):
        self.vs.check(self.lib.vs_stab_sync(self.h), self.h)

    def clean(self):
        self.vs.check(self.lib.vs_stab_clean(self.h), self.h)

    def set_batch(self, frames):
        self.vs.check(self.lib.vs_stab_set_batch(self.h, int(frames)), self.h)

    def set_nv12_layout(self, in_uv_offset=0, out_uv_offset=0):
        self.vs.check(self.lib.vs_stab_set_nv12_layout(self.h, in_uv_offset, out_uv_offset), self.h)

    def set_zero_copy(self, on=True):
        self.vs.check(self.lib.vs_stab_set_zero_copy(self.h, int(on)), self.h)

    def set_warp_batch(self, frames):
        self.vs.check(self.lib.vs_stab_set_warp_batch(self.h, int(frames)), self.h)

    def set_profiling(self, mode):
        self.vs.check(self.lib.vs_stab_set_profiling(self.h, int(mode)), self.h)

    def stage_times(self):
        """(total_ms[STAGE_COUNT], launches[STAGE_COUNT]) accumulated since the last call (synchronises)."""
        ms = (C.c_double * STAGE_COUNT)()
        n = (C.c_int64 * STAGE_COUNT)()
        self.vs.check(self.lib.vs_stab_get_stage_times(self.h, ms, n), self.h)
        return list(ms), list(n)

    def counters(self):
        c = VsCounters()
        self.vs.check(self.lib.vs_stab_get_counters(self.h, C.byref(c)), self.h)
        return c

    def debug(self):
        d = VsDebugFrame()
        self.vs.check(self.lib.vs_stab_get_debug(self.h, C.byref(d)), self.h)
        return d

    def canvas_info(self):
        info = np.zeros(8, np.int32)
        self.vs.check(self.lib.vs_stab_canvas_info(self.h, info.ctypes.data_as(C.POINTER(C.c_int32))), self.h)
        return info

    def debug_arrays(self):
        d = self.debug()
        prev = np.zeros((max(d.n_prev, 1), 2), np.float32)
        cur = np.zeros((max(d.n_prev, 1), 2), np.float32)
        st = np.zeros(max(d.n_prev, 1), np.uint8)
        inl = np.zeros(max(d.n_valid, 1), np.uint8)
        det = np.zeros((max(d.n_detected, 1), 2), np.float32)
        gray = np.zeros(1920 * 1080, np.uint8)
        aw, ah = C.c_int32(), C.c_int32()
        self.vs.check(self.lib.vs_stab_get_debug_arrays(self.h, _p(prev, f32p), _p(cur, f32p), _p(st, u8p),
                                                        _p(inl, u8p), _p(det, f32p), _p(gray, u8p),
                                                        C.byref(aw), C.byref(ah)), self.h)
        return dict(prev=prev[:d.n_prev], curr=cur[:d.n_prev], status=st[:d.n_prev], inliers=inl[:d.n_valid],
                    detected=det[:d.n_detected] if d.detected else det[:0],
                    gray=gray[:aw.value * ah.value].reshape(ah.value, aw.value))


class Batch:
    """vs_batch: n_streams streams of one device scheduled together (one launch per stage over the frames of all of them)."""

    def __init__(self, vs, params, n_streams, frames_per_step, device=0):
        """params: one VsParams for all streams, or a list of n_streams of them (vs_batch_create_params)."""
        self.vs, self.lib, self.n = vs, vs.lib, n_streams
        h = C.c_void_p()
        if isinstance(params, (list, tuple)):
            assert len(params) == n_streams
            arr = (VsParams * n_streams)(*params)
            vs.check(self.lib.vs_batch_create_params(device, n_streams, arr, frames_per_step, C.byref(h)))
        else:
            vs.check(self.lib.vs_batch_create(device, n_streams, C.byref(params), frames_per_step, C.byref(h)))
        self.h = h
        self._produced = (C.c_int32 * n_streams)()

    def _check(self, status):
        if status != 0:
            msg = self.lib.vs_status_string(status).decode()
            raise VsError("%s: %s" % (msg, (self.lib.vs_batch_last_error(self.h) or b"").decode()))

    def close(self):
        if self.h:
            self.lib.vs_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self, i):
        """Member i as a Stabilizer view (getters only; it is not closed through the view)."""
        s = Stabilizer.__new__(Stabilizer)
        s.vs, s.lib = self.vs, self.lib
        s.h = C.c_void_p(self.lib.vs_batch_stream(self.h, i))
        s.close = lambda: None
        return s

    def set_zero_copy(self, on=True):
        self._check(self.lib.vs_batch_set_zero_copy(self.h, int(on)))

    def set_nv12_layout(self, in_uv_offset=0, out_uv_offset=0):
        self._check(self.lib.vs_batch_set_nv12_layout(self.h, in_uv_offset, out_uv_offset))

    def push_dev(self, d_frames, w, h, stride, fmt, d_outs, out_stride):
        """d_frames / d_outs: one device pointer per stream (None: no frame for that stream).  Returns produced[]."""
        fr = (C.c_void_p * self.n)(*[C.c_void_p(p) if p else None for p in d_frames])
        ou = (C.c_void_p * self.n)(*[C.c_void_p(p) if p else None for p in d_outs])
        self._check(self.lib.vs_batch_push_dev(self.h, fr, w, h, stride, fmt, ou, out_stride, self._produced))
        return list(self._produced)

    def pointer_array(self, ptrs):
        """A reusable argument of push_dev_arrays: one device pointer per stream."""
        return (C.c_void_p * self.n)(*[C.c_void_p(p) if p else None for p in ptrs])

    def push_dev_arrays(self, fr, w, h, stride, fmt, ou, out_stride):
        """push_dev with prepared pointer arrays (pointer_array): no per-call marshalling."""
        self._check(self.lib.vs_batch_push_dev(self.h, fr, w, h, stride, fmt, ou, out_stride, self._produced))

    def flush_dev(self, d_outs, out_stride):
        ou = (C.c_void_p * self.n)(*[C.c_void_p(p) if p else None for p in d_outs])
        self._check(self.lib.vs_batch_flush_dev(self.h, ou, out_stride, self._produced))
        return list(self._produced)

    def sync(self):
        self._check(self.lib.vs_batch_sync(self.h))


_cached = None


def load(path=None):
    """Load libvideo-stab.so; raises if it has not been built (no fallback)."""
    global _cached
    if _cached is None:
        # (VS_LAB=1 VS_LIB_PATH=...: a measurement build of the library, scratch/blend_lab.sh)
        path = path or (os.environ.get("VS_LIB_PATH") if os.environ.get("VS_LAB") == "1" else None) or LIB_PATH
        if not os.path.exists(path):
            raise VsError("libvideo-stab.so not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _cached = VsLib(C.CDLL(path))
    return _cached
