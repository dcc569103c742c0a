"""ctypes binding of the CPU oracle (oracle/_build/libvso_oracle.so).

Test infrastructure: imported only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# VSO_ORACLE_LIB: another build of the oracle (the sanitizer build of `make asan`)
LIB = os.environ.get("VSO_ORACLE_LIB") or os.path.join(ORACLE_DIR, "_build", "libvso_oracle.so")

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
i16p = C.POINTER(C.c_int16)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def _p(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    def __init__(self, lib):
        from vsamd.capi import VsParams, VsDebugFrame
        self.lib = lib
        self.VsParams = VsParams
        self.VsDebugFrame = VsDebugFrame
        lib.vso_stab_create.restype = C.c_void_p
        lib.vso_stab_create.argtypes = [C.POINTER(VsParams)]
        lib.vso_stab_destroy.argtypes = [C.c_void_p]
        lib.vso_stab_clean.argtypes = [C.c_void_p]
        lib.vso_stab_push.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, u8p, C.c_size_t]
        lib.vso_stab_flush.argtypes = [C.c_void_p, u8p, C.c_size_t]
        lib.vso_stab_out_size.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, i32p]
        lib.vso_stab_get_debug.argtypes = [C.c_void_p, C.POINTER(VsDebugFrame)]
        lib.vso_stab_canvas_info.argtypes = [C.c_void_p, i32p]
        lib.vso_stab_get_debug_arrays.argtypes = [C.c_void_p, f32p, f32p, u8p, u8p, f32p, u8p, i32p, i32p]
        lib.vso_params_default.argtypes = [C.POINTER(VsParams)]
        lib.vso_resize_linear_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, u8p, C.c_int, C.c_int, C.c_size_t]
        lib.vso_bgr2gray.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t]
        lib.vso_pyr_down.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t]
        lib.vso_scharr.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, i16p]
        lib.vso_copy_make_border.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, u8p, C.c_size_t, C.c_int, C.c_int]
        lib.vso_warp_affine.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, u8p, C.c_size_t, f32p]
        lib.vso_warp_affine_mt.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, u8p, C.c_size_t, f32p, C.c_int]
        lib.vso_warp_affine_nv12.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t, f32p]
        lib.vso_min_eigen.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, f32p]
        lib.vso_gftt.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_double, C.c_double, C.c_int, f32p, i32p]
        lib.vso_pyr_lk.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_size_t, f32p, C.c_int, f32p, u8p, f32p,
                                   C.c_int, C.c_int, C.c_int, C.c_double]
        lib.vso_rng_stream.argtypes = [C.c_uint64, C.POINTER(C.c_uint32), C.c_int]
        lib.vso_estimate_affine_partial2d.argtypes = [f32p, f32p, C.c_int, C.c_double, C.c_int, f64p, u8p, i32p]
        lib.vso_box_filter.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p]
        lib.vso_gaussian_filter.argtypes = [f32p, C.c_int, C.c_float, f32p]
        lib.vso_kalman_filter.argtypes = [f32p, C.c_int, f32p]
        lib.vso_adaptive_radius.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int]
        lib.vso_motion_intent.argtypes = [f32p, C.c_int, C.c_int]
        lib.vso_set_threads.argtypes = [C.c_int]
        lib.vso_libm_checksum.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int]
        lib.vso_libm_checksum.restype = C.c_uint64
        from vsamd.capi import VsRollParams
        self.VsRollParams = VsRollParams
        lib.vso_roll_params_default.argtypes = [C.POINTER(VsRollParams)]
        lib.vso_roll_params_default.restype = None
        lib.vso_sobel16.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, i16p, i16p]
        lib.vso_sobel16.restype = None
        lib.vso_canny.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_double, u8p]
        lib.vso_canny.restype = None
        lib.vso_hough_lines.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_float, C.c_float, C.c_int, f32p, C.c_int]
        lib.vso_warp_affine_d.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, u8p, C.c_size_t, f64p, C.c_int]
        lib.vso_warp_affine_d.restype = None
        lib.vso_content_mask.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, u8p]
        lib.vso_content_mask.restype = None
        lib.vso_find_contours.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, i32p, C.c_int, i32p, C.c_int]
        lib.vso_fill_contour.argtypes = [i32p, C.c_int, C.c_int, C.c_int, u8p]
        lib.vso_fill_contour.restype = None
        lib.vso_azc_crop_rect.argtypes = [u8p, C.c_int, C.c_int, i32p]
        lib.vso_azc_crop_rect.restype = None
        lib.vso_azc_apply.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, u8p, i32p, i32p, i32p]
        lib.vso_roll_create.restype = C.c_void_p
        lib.vso_roll_create.argtypes = [C.POINTER(VsRollParams)]
        lib.vso_roll_destroy.argtypes = [C.c_void_p]
        lib.vso_roll_destroy.restype = None
        lib.vso_roll_correct.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_size_t, u8p, C.c_size_t]
        lib.vso_roll_get.argtypes = [C.c_void_p, f64p, f64p, i32p, i32p]
        lib.vso_roll_get.restype = None

    # ---- primitives -------------------------------------------------------
    def params(self, **kw):
        p = self.VsParams()
        self.lib.vso_params_default(C.byref(p))
        for k, v in kw.items():
            assert hasattr(p, k), k
            setattr(p, k, v)
        return p

    def resize(self, img, dw, dh):
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        out = np.empty((dh, dw) if cn == 1 else (dh, dw, cn), np.uint8)
        self.lib.vso_resize_linear_u8(_p(img, u8p), w, h, w * cn, cn, _p(out, u8p), dw, dh, dw * cn)
        return out

    def bgr2gray(self, img):
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        out = np.empty((h, w), np.uint8)
        self.lib.vso_bgr2gray(_p(img, u8p), w, h, w * 3, _p(out, u8p), w)
        return out

    def analysis_gray(self, frame, aw, ah):
        if frame.ndim == 3:
            return self.bgr2gray(self.resize(frame, aw, ah))
        return self.resize(frame, aw, ah)

    def pyr_down(self, g):
        g = np.ascontiguousarray(g)
        h, w = g.shape
        out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
        self.lib.vso_pyr_down(_p(g, u8p), w, h, w, _p(out, u8p), out.shape[1])
        return out

    def scharr(self, g):
        g = np.ascontiguousarray(g)
        h, w = g.shape
        out = np.empty((h, w, 2), np.int16)
        self.lib.vso_scharr(_p(g, u8p), w, h, w, _p(out, i16p))
        return out

    def copy_make_border(self, img, b, border):
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        out = np.empty((h + 2 * b, w + 2 * b) if cn == 1 else (h + 2 * b, w + 2 * b, cn), np.uint8)
        self.lib.vso_copy_make_border(_p(img, u8p), w, h, w * cn, cn, _p(out, u8p), (w + 2 * b) * cn, b, border)
        return out

    def warp_affine(self, img, M, threads=1):
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        M = np.ascontiguousarray(M, np.float32).reshape(6)
        out = np.empty_like(img)
        self.lib.vso_warp_affine_mt(_p(img, u8p), w, h, w * cn, cn, _p(out, u8p), w * cn, _p(M, f32p), threads)
        return out

    def warp_affine_nv12(self, img, w, h, M):
        img = np.ascontiguousarray(img)
        M = np.ascontiguousarray(M, np.float32).reshape(6)
        out = np.empty_like(img)
        self.lib.vso_warp_affine_nv12(_p(img, u8p), w, h, w, _p(out, u8p), w, _p(M, f32p))
        return out

    def min_eigen(self, g, block_size=3):
        g = np.ascontiguousarray(g)
        h, w = g.shape
        out = np.empty((h, w), np.float32)
        self.lib.vso_min_eigen(_p(g, u8p), w, h, w, block_size, _p(out, f32p))
        return out

    def gftt(self, g, max_corners, quality, min_distance, block_size=3):
        g = np.ascontiguousarray(g)
        h, w = g.shape
        cap = max_corners if max_corners > 0 else w * h
        out = np.zeros((cap, 2), np.float32)
        nc = C.c_int32(0)
        n = self.lib.vso_gftt(_p(g, u8p), w, h, w, max_corners, quality, min_distance, block_size,
                              _p(out, f32p), C.byref(nc))
        return out[:n].copy(), nc.value

    def pyr_lk(self, prev, nxt, pts, win=15, max_level=2, iters=20, eps=0.03):
        prev = np.ascontiguousarray(prev)
        nxt = np.ascontiguousarray(nxt)
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        n = pts.shape[0]
        h, w = prev.shape
        out = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        self.lib.vso_pyr_lk(_p(prev, u8p), _p(nxt, u8p), w, h, w, _p(pts, f32p), n, _p(out, f32p),
                            _p(st, u8p), _p(err, f32p), win, max_level, iters, eps)
        return out, st, err

    def rng_stream(self, seed, n):
        out = np.zeros(n, np.uint32)
        self.lib.vso_rng_stream(seed, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return out

    def estimate_affine_partial2d(self, a, b, thr=5.0, max_iters=500):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(b, np.float32).reshape(-1, 2)
        n = a.shape[0]
        model = np.zeros(6, np.float64)
        inl = np.zeros(max(n, 1), np.uint8)
        info = np.zeros(4, np.int32)
        ok = self.lib.vso_estimate_affine_partial2d(_p(a, f32p), _p(b, f32p), n, thr, max_iters,
                                                     _p(model, f64p), _p(inl, u8p), _p(info, i32p))
        return ok, model, inl[:n], info

    def box_filter(self, path, radius_param, drone=False):
        path = np.ascontiguousarray(path, np.float32)
        out = np.empty_like(path)
        self.lib.vso_box_filter(_p(path, f32p), len(path), radius_param, int(drone), _p(out, f32p))
        return out

    def gaussian_filter(self, path, sigma):
        path = np.ascontiguousarray(path, np.float32)
        out = np.empty_like(path)
        self.lib.vso_gaussian_filter(_p(path, f32p), len(path), sigma, _p(out, f32p))
        return out

    def kalman_filter(self, path):
        path = np.ascontiguousarray(path, np.float32)
        out = np.empty_like(path)
        self.lib.vso_kalman_filter(_p(path, f32p), len(path), _p(out, f32p))
        return out

    def adaptive_radius(self, px, py, pa, smoothing_radius):
        px, py, pa = [np.ascontiguousarray(v, np.float32) for v in (px, py, pa)]
        return self.lib.vso_adaptive_radius(_p(px, f32p), _p(py, f32p), _p(pa, f32p), len(px), smoothing_radius)

    def motion_intent(self, transforms, frame_index):
        t = np.ascontiguousarray(transforms, np.float32).reshape(-1, 3)
        return self.lib.vso_motion_intent(_p(t, f32p), t.shape[0], frame_index)

    def sobel16(self, g):
        g = np.ascontiguousarray(g)
        h, w = g.shape
        dx = np.empty((h, w), np.int16)
        dy = np.empty((h, w), np.int16)
        self.lib.vso_sobel16(_p(g, u8p), w, h, w, _p(dx, i16p), _p(dy, i16p))
        return dx, dy

    def canny(self, g, low, high):
        g = np.ascontiguousarray(g)
        h, w = g.shape
        e = np.empty((h, w), np.uint8)
        self.lib.vso_canny(_p(g, u8p), w, h, w, low, high, _p(e, u8p))
        return e

    def hough_lines(self, edges, rho, theta, threshold, max_lines=65536):
        edges = np.ascontiguousarray(edges)
        h, w = edges.shape
        out = np.empty((max_lines, 2), np.float32)
        n = self.lib.vso_hough_lines(_p(edges, u8p), w, h, w, rho, theta, threshold, _p(out, f32p), max_lines)
        return out[:n].copy()

    def warp_affine_d(self, img, M, border=0):
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        M = np.ascontiguousarray(M, np.float64).reshape(6)
        out = np.empty_like(img)
        self.lib.vso_warp_affine_d(_p(img, u8p), w, h, w * cn, cn, _p(out, u8p), w * cn, _p(M, f64p), border)
        return out

    def content_mask(self, img):
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        cn = 1 if img.ndim == 2 else 3
        m = np.empty((h, w), np.uint8)
        self.lib.vso_content_mask(_p(img, u8p), w, h, w * cn, cn, _p(m, u8p))
        return m

    def find_contours(self, mask):
        """cv::findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) -> list of (n,2) int arrays."""
        mask = np.ascontiguousarray(mask)
        h, w = mask.shape
        counts = np.zeros(w * h // 2 + 4, np.int32)
        xy = np.zeros((2 * w * h + 16, 2), np.int32)
        n = self.lib.vso_find_contours(_p(mask, u8p), w, h, w, _p(counts, i32p), len(counts), _p(xy, i32p), len(xy))
        assert n >= 0
        out, k = [], 0
        for i in range(n):
            out.append(xy[k:k + counts[i]].copy())
            k += counts[i]
        return out

    def fill_contour(self, pts, w, h):
        pts = np.ascontiguousarray(pts, np.int32).reshape(-1, 2)
        m = np.empty((h, w), np.uint8)
        self.lib.vso_fill_contour(_p(pts, i32p), len(pts), w, h, _p(m, u8p))
        return m

    def azc_crop_rect(self, cmask):
        cmask = np.ascontiguousarray(cmask)
        h, w = cmask.shape
        info = np.zeros(8, np.int32)
        self.lib.vso_azc_crop_rect(_p(cmask, u8p), w, h, _p(info, i32p))
        return info

    def auto_zoom_crop(self, frame):
        frame = np.ascontiguousarray(frame)
        h, w = frame.shape[:2]
        cn = 1 if frame.ndim == 2 else 3
        buf = np.empty(max(w * h, 640 * 360) * cn, np.uint8)
        ow, oh = C.c_int32(), C.c_int32()
        info = np.zeros(8, np.int32)
        self.lib.vso_azc_apply(_p(frame, u8p), w, h, w * cn, cn, _p(buf, u8p), C.byref(ow), C.byref(oh), _p(info, i32p))
        shape = (oh.value, ow.value) if cn == 1 else (oh.value, ow.value, 3)
        return buf[:oh.value * ow.value * cn].reshape(shape).copy(), info

    def auto_zoom_crop_nv12(self, surf, w, h):
        """surf: (h * 3 / 2, w) NV12 surface -> the (360 * 3 / 2, 640) result (or the unchanged surface on the fall-back paths), info."""
        surf = np.ascontiguousarray(surf)
        ow_max, oh_max = max(w, 640), max(h, 360)
        buf = np.zeros((oh_max * 3 // 2, ow_max), np.uint8)
        ow, oh = C.c_int32(), C.c_int32()
        info = np.zeros(8, np.int32)
        self.lib.vso_azc_apply_nv12.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, u8p, C.c_size_t, C.c_size_t, i32p, i32p, i32p]
        self.lib.vso_azc_apply_nv12(_p(surf, u8p), w, h, w, w * h, _p(buf, u8p), ow_max, ow_max * oh_max, C.byref(ow), C.byref(oh), _p(info, i32p))
        W, H = ow.value, oh.value
        out = np.concatenate([buf[:H, :W], buf[oh_max:oh_max + H // 2, :W]])
        return out, info

    # ---- enhancer (vso_enhance.cpp) ---------------------------------------
    def enh_params(self, **kw):
        from vsamd.capi import VsEnhParams
        p = VsEnhParams()
        self.lib.vso_enh_params_default.restype = None
        self.lib.vso_enh_params_default(C.byref(p))
        for k, v in kw.items():
            assert hasattr(p, k), k
            setattr(p, k, v)
        return p

    def convert_scale_lut(self, alpha, beta):
        lut = np.empty(256, np.uint8)
        self.lib.vso_convert_scale_lut.restype = None
        self.lib.vso_convert_scale_lut.argtypes = [C.c_double, C.c_double, u8p]
        self.lib.vso_convert_scale_lut(alpha, beta, _p(lut, u8p))
        return lut

    def wb_scales(self, sums, npix, alpha):
        sums = np.ascontiguousarray(sums, np.uint64)
        out = np.empty(3, np.float64)
        self.lib.vso_wb_scales.restype = None
        self.lib.vso_wb_scales.argtypes = [C.c_void_p, C.c_uint64, C.c_float, f64p]
        self.lib.vso_wb_scales(sums.ctypes.data, npix, alpha, _p(out, f64p))
        return out

    def gamma_lut(self, gamma):
        lut = np.empty(256, np.uint8)
        self.lib.vso_gamma_lut.restype = None
        self.lib.vso_gamma_lut.argtypes = [C.c_float, u8p]
        self.lib.vso_gamma_lut(gamma, _p(lut, u8p))
        return lut

    def cvt_color(self, code, px):
        px = np.ascontiguousarray(px, np.uint8).reshape(-1, 3)
        out = np.empty_like(px)
        f = getattr(self.lib, "vso_" + code)
        f.restype = None
        f.argtypes = [u8p, C.c_size_t, u8p]
        f(_p(px, u8p), len(px), _p(out, u8p))
        return out

    def vibrance(self, px, alpha):
        px = np.ascontiguousarray(px, np.uint8).reshape(-1, 3).copy()
        self.lib.vso_vibrance.restype = None
        self.lib.vso_vibrance.argtypes = [u8p, C.c_size_t, C.c_float]
        self.lib.vso_vibrance(_p(px, u8p), len(px), alpha)
        return px

    def gaussian_kernel_q8(self, sigma):
        k = np.zeros(256, np.uint16)
        self.lib.vso_gaussian_kernel_q8.argtypes = [C.c_double, C.c_void_p, C.c_int]
        n = self.lib.vso_gaussian_kernel_q8(sigma, k.ctypes.data, 256)
        return k[:max(n, 0)].copy()

    def gaussian_blur(self, img, sigma):
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        out = np.empty_like(img)
        self.lib.vso_gaussian_blur_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_double, u8p, C.c_size_t]
        n = self.lib.vso_gaussian_blur_u8(_p(img, u8p), w, h, w * cn, cn, sigma, _p(out, u8p), w * cn)
        assert n > 0
        return out

    def add_weighted(self, a, alpha, b, beta, gamma=0.0):
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
        out = np.empty_like(a)
        self.lib.vso_add_weighted_u8.restype = None
        self.lib.vso_add_weighted_u8.argtypes = [u8p, C.c_double, u8p, C.c_double, C.c_double, u8p, C.c_size_t]
        self.lib.vso_add_weighted_u8(_p(a, u8p), alpha, _p(b, u8p), beta, gamma, _p(out, u8p), a.size)
        return out

    def clahe(self, plane, clip_limit, tiles, want_lut=False):
        plane = np.ascontiguousarray(plane, np.uint8)
        h, w = plane.shape
        out = np.empty_like(plane)
        lut = np.zeros((tiles * tiles, 256), np.uint8)
        self.lib.vso_clahe_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_int, u8p, C.c_size_t, u8p]
        rc = self.lib.vso_clahe_u8(_p(plane, u8p), w, h, w, clip_limit, tiles, _p(out, u8p), w, _p(lut, u8p))
        assert rc == 0
        return (out, lut) if want_lut else out

    def nlm_weights(self, h, cn, template=7, search=21):
        info = np.zeros(2, np.int32)
        self.lib.vso_nlm_weights.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, i32p]
        n = self.lib.vso_nlm_weights(h, cn, template, search, None, 0, _p(info, i32p))
        tab = np.zeros(n, np.int32)
        self.lib.vso_nlm_weights(h, cn, template, search, tab.ctypes.data, n, _p(info, i32p))
        return tab, int(info[0]), int(info[1])

    def fast_nl_means(self, img, h, template=7, search=21):
        img = np.ascontiguousarray(img, np.uint8)
        hh, ww = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        out = np.empty_like(img)
        self.lib.vso_fast_nl_means.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_float, C.c_int, C.c_int, u8p, C.c_size_t]
        rc = self.lib.vso_fast_nl_means(_p(img, u8p), ww, hh, ww * cn, cn, h, template, search, _p(out, u8p), ww * cn)
        assert rc == 0
        return out

    def denoise_colored(self, img, h, hc):
        img = np.ascontiguousarray(img, np.uint8).copy()
        hh, ww = img.shape[:2]
        self.lib.vso_denoise_colored.argtypes = [u8p, C.c_int, C.c_int, C.c_float, C.c_float]
        rc = self.lib.vso_denoise_colored(_p(img, u8p), ww, hh, h, hc)
        assert rc == 0
        return img

    def enhance(self, frame, params):
        frame = np.ascontiguousarray(frame)
        h, w = frame.shape[:2]
        out = np.empty_like(frame)
        self.lib.vso_enhance.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, u8p, C.c_size_t]
        rc = self.lib.vso_enhance(_p(frame, u8p), w, h, w * 3, C.cast(C.byref(params), C.c_void_p), _p(out, u8p), w * 3)
        if rc != 0:
            raise RuntimeError("vso_enhance: %d" % rc)
        return out

    def roll_params(self, **kw):
        p = self.VsRollParams()
        self.lib.vso_roll_params_default(C.byref(p))
        for k, v in kw.items():
            assert hasattr(p, k), k
            setattr(p, k, v)
        return p

    def roll_correction(self, params=None):
        return OracleRoll(self, params or self.roll_params())

    def stabilizer(self, params):
        return OracleStab(self, params)


class OracleRoll:
    """vs::RollCorrection::autoCorrectRoll restated on the CPU (oracle/vso_roll.cpp)."""

    def __init__(self, o, params):
        self.lib = o.lib
        self.h = self.lib.vso_roll_create(C.byref(params))

    def close(self):
        if self.h:
            self.lib.vso_roll_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def correct(self, frame):
        frame = np.ascontiguousarray(frame)
        h, w = frame.shape[:2]
        out = np.empty_like(frame)
        self.lib.vso_roll_correct(self.h, _p(frame, u8p), w, h, w * 3, _p(out, u8p), w * 3)
        return out

    def correct_nv12(self, surf, w, h):
        """surf: (h * 3 / 2, w) NV12 surface, packed."""
        surf = np.ascontiguousarray(surf)
        out = np.empty_like(surf)
        self.lib.vso_roll_correct_nv12.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, u8p, C.c_size_t, C.c_size_t]
        self.lib.vso_roll_correct_nv12(self.h, _p(surf, u8p), w, h, w, w * h, _p(out, u8p), w, w * h)
        return out

    def state(self):
        s, d = C.c_double(), C.c_double()
        n, u = C.c_int32(), C.c_int32()
        self.lib.vso_roll_get(self.h, C.byref(s), C.byref(d), C.byref(n), C.byref(u))
        return s.value, d.value, n.value, u.value


class OracleStab:
    """vs::Stabilizer restated on the CPU (oracle/vso_stabilizer.cpp)."""

    def __init__(self, o, params):
        self.o = o
        self.lib = o.lib
        self.h = self.lib.vso_stab_create(C.byref(params))

    def close(self):
        if self.h:
            self.lib.vso_stab_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def clean(self):
        self.lib.vso_stab_clean(self.h)

    def canvas_info(self):
        info = np.zeros(8, np.int32)
        self.lib.vso_stab_canvas_info(self.h, info.ctypes.data_as(C.POINTER(C.c_int32)))
        return info

    def out_shape(self, frame, fmt):
        w = frame.shape[1]
        h = frame.shape[0] if fmt != 1 else frame.shape[0] * 2 // 3
        ow, oh = C.c_int32(), C.c_int32()
        self.lib.vso_stab_out_size(self.h, w, h, C.byref(ow), C.byref(oh))
        if fmt == 0:
            return (oh.value, ow.value, 3)
        if fmt == 1:
            return (oh.value * 3 // 2, ow.value)
        return (oh.value, ow.value)

    def push(self, frame, fmt=0):
        frame = np.ascontiguousarray(frame)
        w = frame.shape[1]
        h = frame.shape[0] if fmt != 1 else frame.shape[0] * 2 // 3
        cn = 3 if fmt == 0 else 1
        out = np.zeros(self.out_shape(frame, fmt), np.uint8)
        r = self.lib.vso_stab_push(self.h, _p(frame, u8p), w, h, w * cn, fmt, _p(out, u8p), out.shape[1] * cn)
        return out if r else None

    def flush(self, like, fmt=0):
        cn = 3 if fmt == 0 else 1
        out = np.zeros(self.out_shape(like, fmt), np.uint8)
        r = self.lib.vso_stab_flush(self.h, _p(out, u8p), out.shape[1] * cn)
        return out if r else None

    def debug(self):
        d = self.o.VsDebugFrame()
        self.lib.vso_stab_get_debug(self.h, C.byref(d))
        return d

    def debug_arrays(self):
        d = self.debug()
        prev = np.zeros((max(d.n_prev, 1), 2), np.float32)
        cur = np.zeros((max(d.n_prev, 1), 2), np.float32)
        st = np.zeros(max(d.n_prev, 1), np.uint8)
        inl = np.zeros(max(d.n_valid, 1), np.uint8)
        det = np.zeros((max(d.n_detected, 1), 2), np.float32)
        gray = np.zeros(960 * 540, np.uint8)
        aw, ah = C.c_int32(), C.c_int32()
        self.lib.vso_stab_get_debug_arrays(self.h, _p(prev, f32p), _p(cur, f32p), _p(st, u8p), _p(inl, u8p),
                                           _p(det, f32p), _p(gray, u8p), C.byref(aw), C.byref(ah))
        return dict(prev=prev[:d.n_prev], curr=cur[:d.n_prev], status=st[:d.n_prev], inliers=inl[:d.n_valid],
                    detected=det[:d.n_detected] if d.detected else det[:0],
                    gray=gray[:aw.value * ah.value].reshape(ah.value, aw.value))


_cached = None


def load():
    global _cached
    if _cached is None:
        if not os.path.exists(LIB):
            build()
        _cached = Oracle(C.CDLL(LIB))
    return _cached
