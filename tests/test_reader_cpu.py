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
    for bad, why in ((b"GIF89a" + bytes(40), "unknown image format"), (jpeg_bytes(img, quality=80, progressive=True), "baseline"),
                     (b"\x89PNG\r\n\x1a\n" + bytes(30), "PNG"), (b"P5\n10 10\n65535\n" + bytes(200), "maxval")):
        with pytest.raises(ValueError, match=why):
            capi.decode_gray8(bad)
    b16 = io.BytesIO()
    Image.fromarray((img.astype(np.uint16) << 8)).save(b16, "PNG")
    with pytest.raises(ValueError, match="8-bit"):
        capi.decode_gray8(b16.getvalue())
    rgb = io.BytesIO()
    Image.fromarray(np.stack([img] * 3, -1)).save(rgb, "PNG")
    with pytest.raises(ValueError, match="grayscale"):
        capi.decode_gray8(rgb.getvalue())


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
