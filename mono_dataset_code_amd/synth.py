"""Synthetic TUM-monoVO-style calibration and frames (no dataset ships with the
reference and none exists in this image).

Writes the files the reference's DatasetReader expects in a sequence folder
(src/BenchmarkDatasetReader.h:86-148): camera.txt (4 lines, src/FOVUndistorter.cpp:63-123),
pcalib.txt (one line of 256 floats, src/PhotometricUndistorter.cpp:70-79),
vignette.png (8/16-bit gray, :120-147), times.txt (:282-323).
"""
import os
import struct
import zlib

import numpy as np

# Representative narrow-lens calibration in the TUM monoVO format (not from the
# reference repository): 1280x1024 FOV camera rectified to a 640x480 pinhole.
CAMERA_1280_TO_640 = ("0.349153 0.436593 0.493140 0.499021 0.933271", "1280 1024", "0.4 0.53 0.5 0.5 0", "640 480")
SEED = 12345


def camera_lines(in_w=1280, in_h=1024, out_w=640, out_h=480, mode="explicit", omega=0.933271):
    l1 = "0.349153 0.436593 0.493140 0.499021 %.6f" % omega
    l3 = {"explicit": "0.4 0.53 0.5 0.5 0", "crop": "crop", "full": "full", "none": "none"}[mode]
    return (l1, "%d %d" % (in_w, in_h), l3, "%d %d" % (out_w, out_h))


def write_camera(path, lines):
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def response_values():
    """A strictly increasing inverse response: gamma-2.2-like curve plus a linear toe."""
    i = np.arange(256, dtype=np.float64)
    return 255.0 * (i / 255.0) ** 2.2 + 0.01 * i + 3.0


def write_pcalib(path, values=None):
    v = response_values() if values is None else values
    with open(path, "w") as f:
        f.write(" ".join("%.9g" % x for x in v) + "\n")


def vignette_image(w, h, bits=16, zeros=3):
    """Radial cos^4-like falloff; a few exact-zero pixels (-> vignetteMapInv = inf)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    r2 = ((x - 0.52 * w) / (0.75 * w)) ** 2 + ((y - 0.48 * h) / (0.75 * w)) ** 2
    v = 1.0 / (1.0 + r2) ** 2
    full = 65535 if bits == 16 else 255
    img = np.clip(np.round(v * full * 0.97), 1, full).astype(np.uint16 if bits == 16 else np.uint8)
    rng = np.random.RandomState(7)
    for _ in range(zeros):
        img[rng.randint(0, h), rng.randint(0, w)] = 0
    return img


def write_png_gray(path, img):
    """Minimal PNG encoder: 8- or 16-bit grayscale, filter 0, one IDAT."""
    assert img.ndim == 2 and img.dtype in (np.uint8, np.uint16)
    h, w = img.shape
    depth = 16 if img.dtype == np.uint16 else 8
    body = img.astype(">u2").tobytes() if depth == 16 else img.tobytes()
    stride = w * depth // 8
    raw = b"".join(b"\x00" + body[y * stride:(y + 1) * stride] for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 0, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


def write_times(path, n):
    with open(path, "w") as f:
        for i in range(n):
            f.write("%05d %.6f %.4f\n" % (i, 1000.0 + i / 20.0, 1.0 + 0.01 * (i % 50)))


def write_sequence_calibration(folder, cam_lines=CAMERA_1280_TO_640, vignette_bits=16, n_times=0):
    """camera.txt + pcalib.txt + vignette.png (+ times.txt) for the given camera lines."""
    os.makedirs(folder, exist_ok=True)
    write_camera(os.path.join(folder, "camera.txt"), cam_lines)
    write_pcalib(os.path.join(folder, "pcalib.txt"))
    w, h = (int(t) for t in cam_lines[1].split())
    write_png_gray(os.path.join(folder, "vignette.png"), vignette_image(w, h, vignette_bits))
    if n_times:
        write_times(os.path.join(folder, "times.txt"), n_times)
    return folder


def _fmix32(h):
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def noise_frames(first_frame, nframes, npix, seed=SEED):
    """Uniform bytes: px(f,i) = fmix32(seed + f*npix + i) >> 24 -- the generator of
    SURVEY.md 8(d); the HIP (mdc_synth_frames_device) and C (orc_synth_frames)
    versions produce the same bytes."""
    idx = (np.arange(first_frame * npix, (first_frame + nframes) * npix, dtype=np.uint64) + np.uint64(seed))
    with np.errstate(over="ignore"):
        return (_fmix32((idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)) >> np.uint32(24)).astype(np.uint8).reshape(nframes, npix)


def smooth_frame(w, h, phase=0.0, blobs=True):
    """Smooth texture with saturated (255) blobs: exercises the NaN mask on regions."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = 127.0 + 100.0 * np.sin(0.01 * x + phase) * np.cos(0.013 * y - phase)
    img = np.clip(np.round(img), 0, 254).astype(np.uint8)
    if blobs:
        for cx, cy, r in ((0.3, 0.4, 0.06), (0.7, 0.55, 0.04), (0.5, 0.5, 0.015)):
            m = (x - cx * w) ** 2 + (y - cy * h) ** 2 < (r * w) ** 2
            img[m] = 255
    return img.reshape(-1)
