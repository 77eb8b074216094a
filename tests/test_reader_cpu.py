"""Host side of the dataset reader (no GPU): the frame decoders against PIL (libjpeg-turbo / libpng), the zip
reader against Python's zipfile, listing / times.txt semantics of class DatasetReader
(reference src/BenchmarkDatasetReader.h:86-148,247-324)."""
import io
import os
import zipfile

import numpy as np
import pytest
from PIL import Image


def c_out(capfd):
    """stdout so far, including what the C++ side printf'ed (flush libc's buffer first)."""
    import ctypes

    ctypes.CDLL(None).fflush(None)
    return capfd.readouterr().out


def textured(h, w, seed=0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    return np.clip(127 + 100 * np.sin(0.05 * x) * np.cos(0.07 * y) + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)


def jpeg_bytes(img, **kw):
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


@pytest.mark.parametrize("size", [(64, 64), (100, 130), (17, 23), (8, 8), (250, 322), (1024, 1280)])
def test_baseline_jpeg_decoder_equals_libjpeg_bit_for_bit(size):
    """Own Huffman decoder + libjpeg's islow inverse DCT: the bytes OpenCV's imread (libjpeg) hands the reference."""
    from mono_dataset_code_amd import capi

    img = textured(*size)
    for kw in ({"quality": 30}, {"quality": 75}, {"quality": 95, "optimize": True}, {"quality": 100}, {"quality": 80, "restart_marker_blocks": 7},
               {"quality": 80, "restart_marker_rows": 1}):
        data = jpeg_bytes(img, **kw)
        assert np.array_equal(capi.decode_gray8(data), np.asarray(Image.open(io.BytesIO(data)))), kw
    # YCbCr files: the luma plane (libjpeg's JCS_GRAYSCALE output), any chroma subsampling
    rgb = np.stack([img, np.roll(img, 3, 1), 255 - img], -1)
    for sub in (0, 1, 2):
        data = jpeg_bytes(rgb, quality=85, subsampling=sub)
        im = Image.open(io.BytesIO(data))
        im.draft("L", im.size)
        assert np.array_equal(capi.decode_gray8(data), np.asarray(im)), sub


def test_png_and_pgm_decoders():
    from mono_dataset_code_amd import capi

    img = textured(77, 131, 3)
    b = io.BytesIO()
    Image.fromarray(img).save(b, "PNG")
    assert np.array_equal(capi.decode_gray8(b.getvalue()), img)
    pgm = b"P5\n# a comment\n131 77\n255\n" + img.tobytes()
    assert np.array_equal(capi.decode_gray8(pgm), img)
    pgm16 = b"P5\n131 77\n65535\n" + (img.astype(">u2") * 257).tobytes()
    assert np.array_equal(capi.decode_gray8(pgm16), img)  # the high byte
    for bad, why in ((b"GIF89a" + bytes(40), "unknown image format"), (b"\x89PNG\r\n\x1a\n" + bytes(30), "PNG"),
                     (b"P5\n10 10\n1000\n" + bytes(200), "maxval")):
        with pytest.raises(ValueError, match=why):
            capi.decode_gray8(bad)


def opencv_gray_of_rgb(rgb):
    """What OpenCV's grayscale read makes of an RGB PNG: libpng's png_set_rgb_to_gray(1, 0.299, 0.587) -- 15-bit fixed
    point, coefficients truncated to 9797 / 19234 / 3737, no rounding for 8-bit samples, + 16384 for 16-bit ones, equal
    channels pass through; 16-bit results keep their high byte."""
    r, g, b = (rgb[..., k].astype(np.uint32) for k in range(3))
    if rgb.dtype == np.uint8:
        y = (9797 * r + 19234 * g + 3737 * b) >> 15
    else:
        y = (9797 * r + 19234 * g + 3737 * b + 16384) >> 15
    y = np.where((r == g) & (r == b), r, y)
    return (y >> 8 if rgb.dtype == np.uint16 else y).astype(np.uint8)


def png_bytes(arr_or_image, **kw):
    b = io.BytesIO()
    (arr_or_image if isinstance(arr_or_image, Image.Image) else Image.fromarray(arr_or_image)).save(b, "PNG", **kw)
    return b.getvalue()


def test_every_png_flavour_opencv_reads_as_gray():
    """cv::imread(..., CV_LOAD_IMAGE_GRAYSCALE) (reference src/BenchmarkDatasetReader.h:252,274) takes any PNG: 16-bit,
    colour, alpha, palette, low bit depths, interlaced.  Gray / alpha / depth cases are compared with PIL's decode of
    the same file (pure sample selection), colour against the libpng formula OpenCV configures (stated above; PIL's own
    'L' conversion rounds differently: the two are shown to differ by at most 1)."""
    from mono_dataset_code_amd import capi

    img = textured(61, 83, 5)
    rng = np.random.default_rng(2)
    rgb = np.stack([img, np.roll(img, 5, 1), (255 - img // 2).astype(np.uint8)], -1)
    rgb[:7] = img[:7, :, None]  # rows with R == G == B
    # 16-bit gray -> high byte
    g16 = (img.astype(np.uint16) << 8) | rng.integers(0, 256, img.shape).astype(np.uint16)
    assert np.array_equal(capi.decode_gray8(png_bytes(g16)), (g16 >> 8).astype(np.uint8))
    # gray + alpha (8 and 16 bit): alpha dropped
    la = np.stack([img, rng.integers(0, 256, img.shape).astype(np.uint8)], -1)
    assert np.array_equal(capi.decode_gray8(png_bytes(Image.fromarray(la, "LA"))), img)
    # RGB, RGBA
    want = opencv_gray_of_rgb(rgb)
    assert np.array_equal(capi.decode_gray8(png_bytes(rgb)), want)
    rgba = np.concatenate([rgb, rng.integers(0, 256, img.shape + (1,)).astype(np.uint8)], -1)
    assert np.array_equal(capi.decode_gray8(png_bytes(Image.fromarray(rgba, "RGBA"))), want)
    pil_l = np.asarray(Image.fromarray(rgb).convert("L")).astype(int)
    assert np.abs(pil_l - want.astype(int)).max() <= 1  # the documented difference to PIL's rounding
    # palette (8-bit and 4-bit indices) -> RGB -> gray
    pal = Image.fromarray(rgb).quantize(colors=200)
    assert np.array_equal(capi.decode_gray8(png_bytes(pal)), opencv_gray_of_rgb(np.asarray(pal.convert("RGB"))))
    pal16 = Image.fromarray(rgb).quantize(colors=13)
    data = png_bytes(pal16, bits=4)
    assert data[24] == 4 and data[25] == 3  # IHDR: depth 4, colour type 3
    assert np.array_equal(capi.decode_gray8(data), opencv_gray_of_rgb(np.asarray(pal16.convert("RGB"))))
    # 1-bit gray -> 0 / 255
    one = Image.fromarray(((img > 127) * 255).astype(np.uint8)).convert("1")
    data = png_bytes(one)
    assert data[24] == 1 and data[25] == 0
    assert np.array_equal(capi.decode_gray8(data), np.asarray(one).astype(np.uint8) * 255)
    # Adam7 interlacing: own minimal encoder (PIL cannot write it), every flavour again, odd sizes incl. 1 x 1 and widths < 8
    for arr in (img, g16, rgb, img[:1, :1], img[:3, :5], rgb[:9, :2], g16[:5, :9]):
        data = interlaced_png(arr)
        want_i = arr if arr.dtype == np.uint8 and arr.ndim == 2 else (arr >> 8).astype(np.uint8) if arr.ndim == 2 else opencv_gray_of_rgb(arr)
        assert np.array_equal(capi.decode_gray8(data), want_i), arr.shape
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(data))).astype(arr.dtype).reshape(arr.shape), arr)  # PIL agrees the file is sound
    # 16-bit RGB (own encoder: PIL does not write it)
    rgb16 = (rgb.astype(np.uint16) << 8) | rng.integers(0, 256, rgb.shape).astype(np.uint16)
    assert np.array_equal(capi.decode_gray8(raw_png(rgb16)), opencv_gray_of_rgb(rgb16))


def _png_chunks(w, h, depth, ctype, interlace, raw):
    import struct
    import zlib

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + chunk(b"IDAT", zlib.compress(raw, 6)) +
            chunk(b"IEND", b""))


def _png_rows(arr):
    """(depth, colour type, list of filtered scanlines) of a gray / RGB, 8 / 16-bit array with filter types cycling 0..4."""
    depth = 16 if arr.dtype == np.uint16 else 8
    ctype = 0 if arr.ndim == 2 else 2
    a = arr.reshape(arr.shape[0], -1)
    body = a.astype(">u2").tobytes() if depth == 16 else a.tobytes()
    stride = a.shape[1] * depth // 8
    bpp = max(1, (1 if arr.ndim == 2 else 3) * depth // 8)
    rows, prev = [], bytes(stride)
    for y in range(arr.shape[0]):
        cur = body[y * stride:(y + 1) * stride]
        ft = y % 5
        out = bytearray(stride)
        for i in range(stride):
            aa = cur[i - bpp] if i >= bpp else 0
            bb = prev[i]
            cc = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                pr = 0
            elif ft == 1:
                pr = aa
            elif ft == 2:
                pr = bb
            elif ft == 3:
                pr = (aa + bb) >> 1
            else:
                p = aa + bb - cc
                pa, pb, pc = abs(p - aa), abs(p - bb), abs(p - cc)
                pr = aa if pa <= pb and pa <= pc else (bb if pb <= pc else cc)
            out[i] = (cur[i] - pr) & 255
        rows.append(bytes([ft]) + bytes(out))
        prev = cur
    return depth, ctype, rows


def raw_png(arr):
    depth, ctype, rows = _png_rows(arr)
    return _png_chunks(arr.shape[1], arr.shape[0], depth, ctype, 0, b"".join(rows))


def interlaced_png(arr):
    """Adam7: the seven reduced images, each filtered on its own."""
    h, w = arr.shape[:2]
    x0, y0, dx, dy = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
    raw, depth, ctype = b"", None, None
    for p in range(7):
        sub = arr[y0[p]::dy[p], x0[p]::dx[p]]
        d, c, rows = _png_rows(sub) if sub.size else (16 if arr.dtype == np.uint16 else 8, 0 if arr.ndim == 2 else 2, [])
        depth, ctype = d, c
        raw += b"".join(rows)
    return _png_chunks(w, h, depth, ctype, 1, raw)


@pytest.mark.parametrize("size", [(64, 64), (100, 130), (17, 23), (8, 8), (250, 322)])
def test_progressive_jpeg_decoder_equals_libjpeg_bit_for_bit(size):
    """SOF2 files (spectral selection + successive approximation): gray and YCbCr with every chroma subsampling, with and
    without restart markers -- the luma plane libjpeg-turbo delivers for JCS_GRAYSCALE, bit for bit."""
    from mono_dataset_code_amd import capi

    img = textured(*size)
    for kw in ({"quality": 30}, {"quality": 85}, {"quality": 95, "optimize": True}, {"quality": 100}, {"quality": 75, "restart_marker_blocks": 5},
               {"quality": 75, "restart_marker_rows": 1}):
        data = jpeg_bytes(img, progressive=True, **kw)
        assert b"\xff\xc2" in data
        assert np.array_equal(capi.decode_gray8(data), np.asarray(Image.open(io.BytesIO(data)))), kw
    rgb = np.stack([img, np.roll(img, 3, 1), 255 - img], -1)
    for sub in (0, 1, 2):
        for extra in ({}, {"restart_marker_blocks": 3}):
            data = jpeg_bytes(rgb, quality=85, subsampling=sub, progressive=True, **extra)
            im = Image.open(io.BytesIO(data))
            im.draft("L", im.size)
            assert np.array_equal(capi.decode_gray8(data), np.asarray(im)), (sub, extra)


def test_vignette_of_any_png_flavour(tmp_path, capfd):
    """cv::imread(vignette, CV_LOAD_IMAGE_UNCHANGED) (reference src/PhotometricUndistorter.cpp:120): every single-channel
    PNG works (interlaced, 16-bit, low bit depth), a multi-channel one is refused LOUDLY (the reference's behaviour for it
    is undefined) instead of silently becoming 1."""
    from mono_dataset_code_amd import capi, synth

    w, h = 48, 32
    v16 = np.asarray(synth.vignette_image(w, h, 16)).reshape(h, w).astype(np.uint16)
    good = " ".join("%.6f" % (255.0 * (i / 255.0) ** 2.2 + 0.01 * i) for i in range(256))
    open(tmp_path / "pcalib.txt", "w").write(good + "\n")
    synth.write_png_gray(str(tmp_path / "plain16.png"), v16)
    open(tmp_path / "interlaced16.png", "wb").write(interlaced_png(v16))
    ref = capi.PhotometricUndistorter(str(tmp_path / "pcalib.txt"), str(tmp_path / "plain16.png"), w, h)
    lace = capi.PhotometricUndistorter(str(tmp_path / "pcalib.txt"), str(tmp_path / "interlaced16.png"), w, h)
    assert ref.valid() == 3 and lace.valid() == 3
    assert np.array_equal(ref.vignette()[1].view(np.uint32), lace.vignette()[1].view(np.uint32))
    c_out(capfd)
    open(tmp_path / "rgb.png", "wb").write(png_bytes(np.stack([(v16 >> 8).astype(np.uint8)] * 3, -1)))
    col = capi.PhotometricUndistorter(str(tmp_path / "pcalib.txt"), str(tmp_path / "rgb.png"), w, h)
    log = c_out(capfd)
    assert col.valid() == 1 and "ERROR: vignette image has 3 channels" in log


def make_sequence(d, frames, zipped=False, fmt="png", times=True, stored=False):
    from mono_dataset_code_amd import synth

    h, w = frames[0].shape
    cam = ("0.349153 0.436593 0.493140 0.499021 0.933271", "%d %d" % (w, h), "crop", "%d %d" % (w * 3 // 5, h * 9 // 16))
    synth.write_sequence_calibration(d, cam, vignette_bits=16, n_times=len(frames) if times else 0)
    names = ["%05d.%s" % (i, fmt) for i in range(len(frames))]
    blobs = []
    for f in frames:
        b = io.BytesIO()
        if fmt == "pgm":
            b.write(b"P5\n%d %d\n255\n" % (w, h) + f.tobytes())
        else:
            Image.fromarray(f).save(b, {"png": "PNG", "jpg": "JPEG"}[fmt], **({"quality": 92} if fmt == "jpg" else {}))
        blobs.append(b.getvalue())
    if zipped:
        with zipfile.ZipFile(os.path.join(d, "images.zip"), "w", zipfile.ZIP_STORED if stored else zipfile.ZIP_DEFLATED) as z:
            for n, b in reversed(list(zip(names, blobs))):  # archive order != name order: the reader sorts (:127)
                z.writestr(n, b)
    else:
        os.makedirs(os.path.join(d, "images"))
        for n, b in zip(names, blobs):
            open(os.path.join(d, "images", n), "wb").write(b)
    return names, blobs


@pytest.mark.parametrize("zipped,fmt,stored", [(False, "png", False), (True, "png", False), (True, "jpg", True), (False, "pgm", False), (True, "pgm", False)])
def test_reader_lists_sorts_decodes(tmp_path, zipped, fmt, stored, capfd):
    from mono_dataset_code_amd import capi

    frames = [textured(64, 80, s) for s in range(7)]
    names, blobs = make_sequence(str(tmp_path), frames, zipped, fmt, stored=stored)
    r = capi.DatasetReader(str(tmp_path))
    log = c_out(capfd)
    if zipped:
        assert "assuming that images are zipped" in log and "got 7 entries and 7 files from zipfile!" in log
    else:
        assert "found 7 files in folder /images" in log
    assert "Got 7 files!" in log
    assert len(r) == 7 and (r.in_w, r.in_h) == (80, 64)
    assert [r.timestamp(i) for i in (0, 3, 6)] == [1000.0, 1000.0 + 3 / 20.0, 1000.0 + 6 / 20.0]
    assert abs(r.exposure(2) - 1.02) < 1e-6 and r.timestamp(-1) == 0 and r.exposure(7) == 0
    for threads, prefetch in ((0, 8), (3, 2), (1, 0)):
        r.set_threads(threads)
        r.set_prefetch(prefetch)
        for i in (0, 1, 2, 6, 3, 3, 5, 4):  # sequential (prefetch hits) and random access
            got = r.get_raw(i)
            want = frames[i] if fmt != "jpg" else np.asarray(Image.open(io.BytesIO(blobs[i])))
            assert got is not None and np.array_equal(got, want), (threads, prefetch, i)
    assert r.get_raw(7) is None and "out of range" in r.last_error()
    r.close()


def test_reader_times_mismatch_and_two_column_times(tmp_path, capfd):
    from mono_dataset_code_amd import capi

    frames = [textured(32, 48, s) for s in range(3)]
    make_sequence(str(tmp_path), frames)
    open(os.path.join(str(tmp_path), "times.txt"), "w").write("0 5.5\n1 6.5\n2 7.5\n")  # no exposure column (:303-307)
    r = capi.DatasetReader(str(tmp_path))
    assert [r.timestamp(i) for i in range(3)] == [5.5, 6.5, 7.5] and r.exposure(1) == 0
    r.close()
    open(os.path.join(str(tmp_path), "times.txt"), "w").write("0 5.5 1.0\n1 6.5 1.0\n")  # one line short (:312-323)
    c_out(capfd)
    r = capi.DatasetReader(str(tmp_path))
    assert "Mismatch between number of images and number of timestamps" in c_out(capfd)
    assert [r.timestamp(i) for i in range(3)] == [0, 0, 0]
    r.close()


def test_reader_wrong_size_frame_and_undecodable_file(tmp_path, capfd):
    from mono_dataset_code_amd import capi

    frames = [textured(32, 48, s) for s in range(3)]
    make_sequence(str(tmp_path), frames)
    Image.fromarray(textured(30, 48)).save(os.path.join(str(tmp_path), "images", "00001.png"))
    open(os.path.join(str(tmp_path), "images", "00002.png"), "wb").write(b"not an image at all")
    r = capi.DatasetReader(str(tmp_path))
    c_out(capfd)
    assert r.get_image(1, 1, 1, 1, 1) is None  # as the reference: size check, message, 0 (:194-199)
    assert "ERROR: expected cv-mat to have dimensions 48 x 32; found 48 x 30" in c_out(capfd)
    assert r.get_image(2, 0, 0, 0, 0) is None
    assert "found 0 x 0" in c_out(capfd)
    r.close()


def test_image_pool_recycles_blocks():
    from mono_dataset_code_amd import capi

    L = capi.host_lib()
    L.mdch_image_pool_trim()
    a = L.mdch_image_alloc(1000)
    L.mdch_image_free(a)
    b = L.mdch_image_alloc(1000)
    # with a GPU the freed page-locked block comes back; without one the pool does not cache (plain heap)
    if L.mdch_image_pool_idle_bytes() == 0 and a == b:
        pass
    L.mdch_image_free(b)
    L.mdch_image_free(b)  # a second free of the same block is ignored, not a crash
    L.mdch_image_pool_trim()
    assert L.mdch_image_pool_idle_bytes() == 0


def test_get_image_raw_internal_is_there_for_opencv_callers(tmp_path):
    """responseCalib's use of the reader (reference src/main_responseCalib.cpp:194): getImageRaw_internal(int) -> cv::Mat, from a
    translation unit that includes only this repo's reader header (tests/dropin/raw_internal.cpp).  No GPU involved."""
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "raw_internal")
    if not os.path.exists(exe):
        pytest.skip("raw_internal not built")
    frames = [textured(48, 64, s) for s in range(5)]
    make_sequence(str(tmp_path), frames, zipped=True, fmt="png")
    out = subprocess.run([exe, str(tmp_path)], stdout=subprocess.PIPE, text=True, timeout=120).stdout
    rows = [l.split() for l in out.splitlines() if l.startswith("RAW ")]
    assert len(rows) == 5
    for i, r in enumerate(rows):
        assert [int(x) for x in r[1:]] == [i, 48, 64, 0, int(frames[i].astype(np.uint64).sum())]


def test_exposure_image_pool_opt_out():
    """MDC_IMAGE_POOL=0: ExposureImage::image is the reference's `new float[w*h]` (src/ExposureImage.h:45,49) -- nothing is kept
    by the pool after a free, and a block the caller swapped in (`new float[]` of its own) is released by the destructor."""
    import subprocess
    import sys

    code = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
from mono_dataset_code_amd import capi
L = capi.host_lib()
L.mdch_image_alloc.restype = ctypes.c_void_p
L.mdch_image_alloc.argtypes = [ctypes.c_ulong]
L.mdch_image_free.argtypes = [ctypes.c_void_p]
L.mdch_image_pool_idle_bytes.restype = ctypes.c_size_t
a = L.mdch_image_alloc(640 * 480)
b = L.mdch_image_alloc(640 * 480)
assert a and b and a != b
ctypes.memset(a, 0x11, 640 * 480 * 4)
L.mdch_image_free(a)
L.mdch_image_free(b)
print("idle", L.mdch_image_pool_idle_bytes())
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env, want_idle in (({"MDC_IMAGE_POOL": "0"}, "idle 0"), ({}, None)):
        e = dict(os.environ)
        e.pop("MDC_IMAGE_POOL", None)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("idle")][-1]
        if want_idle:
            assert line == want_idle
        else:
            assert int(line.split()[1]) >= 2 * 640 * 480 * 4  # the pool keeps freed blocks for the next images
