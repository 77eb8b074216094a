"""TEST INFRASTRUCTURE ONLY -- ctypes access to the CPU oracle.

  Oracle()  -> oracle/liboracle.so      plain-C restatement (oracle/mdc_oracle.c); travels to the GPU box
  Ref()     -> oracle/_ref/libmdc_ref.so the reference's own sources compiled where they lie (oracle/Makefile);
               buildable only where /root/reference is mounted, but the built .so travels

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libmdc_ref.so")
REFERENCE_ROOT = "/root/reference"

_vp, _i = C.c_void_p, C.c_int


def _p(a):
    return a.ctypes.data_as(_vp) if a is not None else None


def build_oracle(force=False):
    src = os.path.join(HERE, "mdc_oracle.c")
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(src) > os.path.getmtime(ORACLE_SO):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"] + (["-B"] if force else []))
    return ORACLE_SO


def build_ref(force=False):
    """Compiles the reference's hot-path sources; returns None where they are not mounted."""
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "src")):
        return REF_SO if os.path.exists(REF_SO) else None
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"] + (["-B"] if force else []))
    return REF_SO


def build_dropin():
    """The playDataset-style test drivers on this repo's own DatasetReader (tests/dropin/*.cpp against
    include/mono_dataset_code/BenchmarkDatasetReader.h): need no reference sources."""
    subprocess.check_call(["make", "-s", "-C", HERE, "dropin"])


def have_ref():
    return os.path.exists(REF_SO)


class Oracle:
    def __init__(self):
        self.L = C.CDLL(build_oracle())
        L = self.L
        L.orc_distort_coordinates.argtypes = [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _i]
        L.orc_fov_setup.argtypes = [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]
        L.orc_parse_camera.argtypes = [C.c_char_p, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
        L.orc_photo_gamma.argtypes = [_vp, _i, _vp, _vp]
        L.orc_parse_pcalib.argtypes = [C.c_char_p, _vp, _i]
        L.orc_photo_vignette.argtypes = [_vp, _i, _i, _vp, _vp]
        L.orc_unmap.argtypes = [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i]
        L.orc_undistort_f32.argtypes = [_vp, _vp, _vp, _vp, _i, _i]
        L.orc_undistort_u8.argtypes = [_vp, _vp, _vp, _vp, _i, _i]
        L.orc_get_image.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i]
        L.orc_pyramid_level.argtypes = [_vp, _i, _i, _vp]
        L.orc_gradients.argtypes = [_vp, _i, _i, _vp, _vp]
        L.orc_gradients.restype = None
        L.orc_vcal_plane_step.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]
        L.orc_vcal_plane_step.restype = None
        L.orc_vcal_vignette_step.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]
        L.orc_vcal_vignette_step.restype = None
        L.orc_vcal_smooth.argtypes = [_vp, _i, _i, _vp, _vp]
        L.orc_vcal_mask_coords.argtypes = [_vp, _vp, _i, _i, _i]
        L.orc_vcal_gradient_mask.argtypes = [_vp, _i, _i, _i]
        L.orc_vcal_gradient_mask.restype = None
        L.orc_vcal_mask_coords.restype = None
        L.orc_vcal_smooth.restype = None
        L.orc_synth_frames.argtypes = [_vp, C.c_longlong, C.c_longlong, _i, C.c_uint]
        L.orc_time_path.argtypes = [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(C.c_double)]
        L.orc_time_path.restype = C.c_double
        for n in ("orc_distort_coordinates", "orc_photo_vignette", "orc_unmap", "orc_undistort_f32",
                  "orc_undistort_u8", "orc_get_image", "orc_pyramid_level", "orc_synth_frames"):
            getattr(L, n).restype = None

    # ---- calibration -----------------------------------------------------------
    def parse_camera(self, path):
        ic, oc = np.zeros(5, np.float32), np.zeros(5, np.float32)
        iw, ih, mode, ow, oh = (C.c_int(0) for _ in range(5))
        ok = self.L.orc_parse_camera(os.fsencode(path), _p(ic), C.byref(iw), C.byref(ih), C.byref(mode), _p(oc),
                                     C.byref(ow), C.byref(oh))
        return {"valid": bool(ok), "in_calib": ic, "in_w": iw.value, "in_h": ih.value, "mode": mode.value,
                "out_calib_in": oc, "out_w": ow.value, "out_h": oh.value}

    def fov_setup(self, cam):
        """cam = parse_camera() result (valid).  Returns the tables of UndistorterFOV's constructor."""
        n = cam["out_w"] * cam["out_h"]
        rx, ry = np.zeros(n, np.float32), np.zeros(n, np.float32)
        oc, kr, ko = np.zeros(5, np.float32), np.zeros(9, np.float32), np.zeros(9, np.float32)
        black = self.L.orc_fov_setup(_p(cam["in_calib"]), cam["in_w"], cam["in_h"], cam["mode"], _p(cam["out_calib_in"]),
                                     cam["out_w"], cam["out_h"], _p(oc), _p(rx), _p(ry), _p(kr), _p(ko))
        return {"remap_x": rx, "remap_y": ry, "out_calib": oc, "K_rect": kr.reshape(3, 3), "K_org": ko.reshape(3, 3),
                "has_black": bool(black)}

    def distort(self, cam, out_calib, x, y):
        self.L.orc_distort_coordinates(_p(cam["in_calib"]), cam["in_w"], cam["in_h"], _p(out_calib), cam["out_w"],
                                       cam["out_h"], _p(x), _p(y), x.size)

    def photo_gamma(self, raw):
        raw = np.ascontiguousarray(raw, np.float32)
        ginv, g = np.zeros(256, np.float32), np.zeros(256, np.float32)
        ok = self.L.orc_photo_gamma(_p(raw), raw.size, _p(ginv), _p(g))
        return (ginv, g) if ok else None

    def parse_pcalib(self, path):
        raw = np.zeros(4096, np.float32)
        n = self.L.orc_parse_pcalib(os.fsencode(path), _p(raw), raw.size)
        return None if n < 0 else raw[:n].copy()

    def photo_vignette(self, px):
        px = np.ascontiguousarray(px)
        assert px.dtype in (np.uint8, np.uint16)
        vm, vi = np.zeros(px.size, np.float32), np.zeros(px.size, np.float32)
        with np.errstate(all="ignore"):
            self.L.orc_photo_vignette(_p(px), 8 if px.dtype == np.uint8 else 16, px.size, _p(vm), _p(vi))
        return vm, vi

    # ---- per-frame --------------------------------------------------------------
    def unmap(self, img, ginv, vinv, valid_gamma, valid_vignette, g, v, o):
        out = np.zeros(img.size, np.float32)
        self.L.orc_unmap(_p(img), _p(out), img.size, _p(ginv), _p(vinv), int(valid_gamma), int(valid_vignette), int(g),
                         int(v), int(o))
        return out

    def undistort(self, img, rx, ry, in_w):
        out = np.zeros(rx.size, np.float32)
        fn = self.L.orc_undistort_f32 if img.dtype == np.float32 else self.L.orc_undistort_u8
        fn(_p(img), _p(out), _p(rx), _p(ry), in_w, rx.size)
        return out

    def get_image(self, raw, in_w, in_h, out_w, out_h, ginv, vinv, valid_gamma, valid_vignette, rx, ry, rectify, g, v, o,
                  out=None):
        have = rx is not None
        n = out_w * out_h if rectify else in_w * in_h
        if out is None:
            out = np.zeros(n, np.float32)
        tmp = np.zeros(in_w * in_h, np.float32)
        self.L.orc_get_image(_p(raw), _p(out), _p(tmp), in_w, in_h, out_w, out_h, _p(ginv), _p(vinv), int(valid_gamma),
                             int(valid_vignette), _p(rx), _p(ry), int(have), int(rectify), int(g), int(v), int(o))
        return out

    def pyramid_level(self, src, w, h):
        dst = np.zeros((w // 2) * (h // 2), np.float32)
        self.L.orc_pyramid_level(_p(np.ascontiguousarray(src, np.float32)), w, h, _p(dst))
        return dst

    def synth_frames(self, first, n, npix, seed):
        out = np.zeros((n, npix), np.uint8)
        self.L.orc_synth_frames(_p(out), first, n, npix, seed)
        return out

    def gradients(self, level, w, h):
        """DSO-style (I, dx, dy) + absSquaredGrad of one level (own definition, parity unpinned)."""
        lv = np.ascontiguousarray(level, np.float32)
        dI, a = np.zeros(3 * w * h, np.float32), np.zeros(w * h, np.float32)
        self.L.orc_gradients(_p(lv), w, h, _p(dI), _p(a))
        return dI.reshape(h * w, 3), a

    def vcal_plane_step(self, images, p2x, p2y, plane_color, vig, oth2):
        """src/main_vignetteCalib.cpp:400-448 -> (new planeColor, FF, FC, E, R); images (n, hI, wI), p2x/p2y (n, np)."""
        n, hI, wI = images.shape
        npnt = p2x.shape[1]
        pc = np.array(plane_color, np.float32, copy=True)
        ff, fc = np.zeros(npnt, np.float32), np.zeros(npnt, np.float32)
        er = np.zeros(2, np.float64)
        self.L.orc_vcal_plane_step(_p(images), _p(p2x), _p(p2y), n, wI, hI, npnt, _p(pc), _p(ff), _p(fc), _p(vig), int(oth2),
                                   er.ctypes.data, er.ctypes.data + 8)
        return pc, ff, fc, float(er[0]), float(er[1])

    def vcal_vignette_step(self, images, p2x, p2y, plane_color, vig, oth2):
        """src/main_vignetteCalib.cpp:455-527 -> (new vignetteFactor, TT, CT, E, R)."""
        n, hI, wI = images.shape
        npnt = p2x.shape[1]
        vf = np.array(vig, np.float32, copy=True)
        tt, ct = np.zeros(hI * wI, np.float32), np.zeros(hI * wI, np.float32)
        er = np.zeros(2, np.float64)
        self.L.orc_vcal_vignette_step(_p(images), _p(p2x), _p(p2y), n, wI, hI, npnt, _p(plane_color), _p(vf), _p(tt), _p(ct), int(oth2),
                                      er.ctypes.data, er.ctypes.data + 8)
        return vf, tt, ct, float(er[0]), float(er[1])

    def vcal_gradient_mask(self, image, max_abs_grad):
        """src/main_vignetteCalib.cpp:293-301 -> the image (h x w) with the masked pixels NaN."""
        a = np.array(image, np.float32, copy=True)
        self.L.orc_vcal_gradient_mask(_p(a), a.shape[1], a.shape[0], int(max_abs_grad))
        return a

    def vcal_mask_coords(self, x, y, wI, hI):
        """src/main_vignetteCalib.cpp:345-357 -> (x, y) with NaN where the plane point falls outside the image."""
        a, b = np.array(x, np.float32, copy=True), np.array(y, np.float32, copy=True)
        self.L.orc_vcal_mask_coords(_p(a), _p(b), a.size, wI, hI)
        return a, b

    def vcal_smooth(self, vig, wI, hI):
        """src/main_vignetteCalib.cpp:541-566 -> (smoothed factors, scratch)."""
        v = np.ascontiguousarray(vig, np.float32)
        tt, ct = np.zeros(hI * wI, np.float32), np.zeros(hI * wI, np.float32)
        self.L.orc_vcal_smooth(_p(v), wI, hI, _p(tt), _p(ct))
        return tt, ct

    def time_path(self, frames, passes, in_w, in_h, out_w, out_h, ginv, vinv, rx, ry, rectify, g, v, o):
        cs = C.c_double(0)
        return self.L.orc_time_path(_p(frames), frames.shape[0], passes, in_w, in_h, out_w, out_h, _p(ginv), _p(vinv),
                                    _p(rx), _p(ry), int(rectify), int(g), int(v), int(o), C.byref(cs))


class Ref:
    """The reference's own classes (RefUndistorterFOV / RefPhotometricUndistorter)."""

    def __init__(self):
        path = build_ref()
        if not path or not os.path.exists(path):
            raise OSError("oracle/_ref/libmdc_ref.so is not built and /root/reference is not mounted")
        self.L = C.CDLL(path)
        L = self.L
        L.ref_fov_create.argtypes = [C.c_char_p]
        L.ref_fov_create.restype = _vp
        L.ref_photo_create.argtypes = [C.c_char_p, C.c_char_p, _i, _i]
        L.ref_photo_create.restype = _vp
        for n in ("ref_fov_destroy", "ref_photo_destroy"):
            getattr(L, n).argtypes = [_vp]
            getattr(L, n).restype = None
        L.ref_fov_valid.argtypes = [_vp]
        L.ref_fov_dims.argtypes = [_vp, _vp]
        L.ref_fov_intrinsics.argtypes = [_vp, _vp]
        L.ref_fov_remap.argtypes = [_vp, _vp, _vp]
        L.ref_fov_distort.argtypes = [_vp, _vp, _vp, _i]
        L.ref_fov_undistort_f32.argtypes = [_vp, _vp, _vp, _i, _i]
        L.ref_fov_undistort_u8.argtypes = [_vp, _vp, _vp, _i, _i]
        L.ref_photo_valid.argtypes = [_vp]
        L.ref_photo_ginv.argtypes = [_vp, _vp]
        L.ref_photo_g.argtypes = [_vp, _vp]
        L.ref_photo_vignette.argtypes = [_vp, _vp, _vp]
        L.ref_photo_unmap.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i]
        L.ref_get_image.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]
        L.ref_time_path.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_double)]
        L.ref_time_path.restype = C.c_double

    def fov(self, camera_txt):
        return RefFov(self.L, camera_txt)

    def photo(self, pcalib, vignette, w, h):
        return RefPhoto(self.L, pcalib, vignette, w, h)

    def get_image(self, fov, photo, raw, rectify, g, v, o, out=None):
        iw, ih, ow, oh = fov.dims()
        if out is None:
            out = np.zeros(ow * oh if rectify else iw * ih, np.float32)
        tmp = np.zeros(iw * ih, np.float32)
        self.L.ref_get_image(fov.h, photo.h, _p(raw), _p(out), _p(tmp), int(rectify), int(g), int(v), int(o))
        return out

    def time_path(self, fov, photo, frames, nthreads, passes, rectify, g, v, o):
        cs = C.c_double(0)
        return self.L.ref_time_path(fov.h, photo.h, _p(frames), frames.shape[0], nthreads, passes, int(rectify), int(g),
                                    int(v), int(o), C.byref(cs))


class RefFov:
    def __init__(self, L, camera_txt):
        self.L = L
        self.h = L.ref_fov_create(os.fsencode(camera_txt))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_fov_destroy(self.h)
            self.h = None

    def is_valid(self):
        return bool(self.L.ref_fov_valid(self.h))

    def dims(self):
        d = np.zeros(4, np.int32)
        self.L.ref_fov_dims(self.h, _p(d))
        return tuple(int(x) for x in d)

    def intrinsics(self):
        o = np.zeros(29, np.float32)
        self.L.ref_fov_intrinsics(self.h, _p(o))
        return {"K_rect": o[0:9].reshape(3, 3).copy(), "K_org": o[9:18].reshape(3, 3).copy(),
                "original": o[18:23].copy(), "omega": float(o[23]), "out_calib": o[24:29].copy()}

    def remap(self):
        if not self.is_valid():
            return None
        _, _, ow, oh = self.dims()
        rx, ry = np.zeros(ow * oh, np.float32), np.zeros(ow * oh, np.float32)
        return (rx, ry) if self.L.ref_fov_remap(self.h, _p(rx), _p(ry)) else None

    def distort_coordinates(self, x, y):
        self.L.ref_fov_distort(self.h, _p(x), _p(y), x.size)

    def undistort(self, img, out):
        fn = self.L.ref_fov_undistort_f32 if img.dtype == np.float32 else self.L.ref_fov_undistort_u8
        fn(self.h, _p(img), _p(out), img.size, out.size)


class RefPhoto:
    def __init__(self, L, pcalib, vignette, w, h):
        self.L, self.w, self.hh = L, w, h
        self.h = L.ref_photo_create(os.fsencode(pcalib), os.fsencode(vignette), w, h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_photo_destroy(self.h)
            self.h = None

    def valid(self):
        return self.L.ref_photo_valid(self.h)

    def ginv(self):
        o = np.zeros(256, np.float32)
        return o if self.L.ref_photo_ginv(self.h, _p(o)) else None

    def g(self):
        o = np.zeros(256, np.float32)
        return o if self.L.ref_photo_g(self.h, _p(o)) else None

    def vignette(self):
        m, i = np.zeros(self.w * self.hh, np.float32), np.zeros(self.w * self.hh, np.float32)
        return (m, i) if self.L.ref_photo_vignette(self.h, _p(m), _p(i)) else None

    def unmap(self, img, out, g, v, o):
        self.L.ref_photo_unmap(self.h, _p(img), _p(out), img.size, int(g), int(v), int(o))


VCAL_REF_SO = os.path.join(HERE, "_ref", "libvcal_ref.so")


class VcalRef:
    """The reference's own vignetteCalib solver loops (oracle/vcal_extract.py + oracle/vcal_ref_wrapper.cpp)."""

    def __init__(self):
        build_ref()
        if not os.path.exists(VCAL_REF_SO):
            raise OSError("oracle/_ref/libvcal_ref.so is not built and /root/reference is not mounted")
        self.L = C.CDLL(VCAL_REF_SO)

    @staticmethod
    def _rows(a):
        return (C.c_void_p * a.shape[0])(*[a[i].ctypes.data for i in range(a.shape[0])])

    def plane_step(self, images, p2x, p2y, gw, gh, plane_color, vig, oth2):
        n, hI, wI = images.shape
        pc = np.array(plane_color, np.float32, copy=True)
        ff, fc = np.zeros(gw * gh, np.float32), np.zeros(gw * gh, np.float32)
        vf = np.array(vig, np.float32, copy=True)
        e, r = C.c_double(0), C.c_double(0)
        self.L.ref_vcal_plane_step(n, self._rows(p2x), self._rows(p2y), self._rows(images.reshape(n, -1)), gw, gh, wI, hI, _p(pc), _p(ff),
                                   _p(fc), _p(vf), int(oth2), C.byref(e), C.byref(r))
        return pc, ff, fc, e.value, r.value

    def gradient_mask(self, image, max_abs_grad):
        a = np.array(image, np.float32, copy=True)
        self.L.ref_vcal_gradient_mask(_p(a), a.shape[1], a.shape[0], int(max_abs_grad))
        return a

    def mask_coords(self, x, y, gw, gh, wI, hI):
        a, b = np.array(x, np.float32, copy=True), np.array(y, np.float32, copy=True)
        self.L.ref_vcal_mask_coords(_p(a), _p(b), gw, gh, wI, hI)
        return a, b

    def smooth(self, vig, wI, hI):
        v = np.array(vig, np.float32, copy=True)
        tt, ct = np.zeros(hI * wI, np.float32), np.zeros(hI * wI, np.float32)
        self.L.ref_vcal_smooth(wI, hI, _p(v), _p(tt), _p(ct))
        return tt, ct

    def vignette_step(self, images, p2x, p2y, gw, gh, plane_color, vig, oth2):
        n, hI, wI = images.shape
        pc = np.array(plane_color, np.float32, copy=True)
        vf = np.array(vig, np.float32, copy=True)
        tt, ct = np.zeros(hI * wI, np.float32), np.zeros(hI * wI, np.float32)
        e, r = C.c_double(0), C.c_double(0)
        self.L.ref_vcal_vignette_step(n, self._rows(p2x), self._rows(p2y), self._rows(images.reshape(n, -1)), gw, gh, wI, hI, _p(pc),
                                      _p(vf), _p(tt), _p(ct), int(oth2), C.byref(e), C.byref(r))
        return vf, tt, ct, e.value, r.value
