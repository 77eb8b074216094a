#!/usr/bin/env python3
"""End-to-end sequence-reader rates on this box (decode + photometric + rectify, 1280x1024 -> 640x480, flags r+g+v+o):
the reference reader as shipped (CPU, one thread), the reference's unmodified reader on this repo's drop-in classes,
this repo's reader frame by frame (getImage) and batched (getImages), from an images/ folder of PNGs, from a zip of
PNGs and from a zip of JPEGs (MDC_RATE_KINDS=folder_png,zip_png,zip_jpg; also zip_jpg_rst: restart intervals, zip_jpg_420 /
zip_jpg_444: colour files).  usage: python tools/reader_rate.py [frames] (default 256)"""
import io
import os
import subprocess
import sys
import tempfile
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

from mono_dataset_code_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BIN = os.path.join(ROOT, "oracle", "_ref")


def textured(seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:1024, 0:1280]
    return np.clip(127 + 100 * np.sin(0.01 * x + seed) * np.cos(0.013 * y) + rng.normal(0, 4, (1024, 1280)), 0, 255).astype(np.uint8)


def make(kind):
    d = tempfile.mkdtemp(prefix="mdc_seq_%s_" % kind)
    synth.write_sequence_calibration(d, synth.CAMERA_1280_TO_640, n_times=N)
    base = [textured(s) for s in range(8)]
    blobs = []
    for i in range(N):
        b = io.BytesIO()
        if kind == "zip_jpg":
            Image.fromarray(base[i % 8]).save(b, "JPEG", quality=90)
        elif kind == "zip_jpg_rst":  # a restart interval per MCU row, as hardware encoders write them
            Image.fromarray(base[i % 8]).save(b, "JPEG", quality=90, restart_marker_rows=1)
        elif kind in ("zip_jpg_420", "zip_jpg_444"):  # colour files (cv::imread(..., GRAYSCALE) keeps the luma plane)
            g = base[i % 8]
            Image.fromarray(np.stack([g, np.roll(g, 5, 1), 255 - g], -1)).save(b, "JPEG", quality=90, subsampling=2 if kind == "zip_jpg_420" else 0)
        else:
            Image.fromarray(base[i % 8]).save(b, "PNG", compress_level=1)
        blobs.append(b.getvalue())
    if kind == "folder_png":
        os.makedirs(os.path.join(d, "images"))
        for i, b in enumerate(blobs):
            open(os.path.join(d, "images", "%05d.png" % i), "wb").write(b)
    else:
        with zipfile.ZipFile(os.path.join(d, "images.zip"), "w", zipfile.ZIP_STORED) as z:
            for i, b in enumerate(blobs):
                z.writestr("%05d.%s" % (i, "jpg" if kind.startswith("zip_jpg") else "png"), b)
    return d, sum(len(b) for b in blobs) / N


def run(binary, folder, passes, *extra, env=None):
    p = os.path.join(BIN, binary)
    if not os.path.exists(p):
        return "%s: not built" % binary
    r = subprocess.run([p, folder, "1111", str(passes)] + list(extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900,
                       env=dict(os.environ, **(env or {})))
    lines = [l for l in r.stdout.splitlines() if l.startswith("READER_RATE")]
    return "\n".join(l.replace(BIN + "/", "") for l in lines) if lines else "%s failed: %s" % (binary, r.stdout[-300:])


ONLY_DEVICE = os.environ.get("MDC_RATE_ONLY") == "device"  # only the device-resident runs (+ MDC_RATE_ENVS: "A=1 B=2;A=3" -> one run per set)
for kind in os.environ.get("MDC_RATE_KINDS", "folder_png,zip_png,zip_jpg").split(","):
    d, avg = make(kind)
    print("== %s: %d frames 1280x1024, %.0f KB/frame on disk" % (kind, N, avg / 1e3), flush=True)
    if ONLY_DEVICE:
        print(run("reader_rate_fast", d, 3, "batch"), flush=True)
        for envs in [""] + [e for e in os.environ.get("MDC_RATE_ENVS", "").split(";") if e]:
            env = dict(kv.split("=", 1) for kv in envs.split())
            for mode in ("device", "device_dso"):
                print("-- getImagesDevice (%s), %s:" % (mode, envs or "defaults"), flush=True)
                print(run("reader_rate_fast", d, 8, mode, env=env), flush=True)
        continue
    if not kind.startswith("zip_jpg"):  # the test shim's imread / imdecode stand-ins decode PNG (libpng), not JPEG
        print(run("reader_rate_ref", d, 1), flush=True)
        print(run("reader_rate_mdc", d, 2), flush=True)
    print(run("reader_rate_fast", d, 3), flush=True)
    if kind.startswith("zip_jpg"):  # getImages: whole decode on the host vs Huffman on the host + inverse DCT on the GPU
        print("-- getImages, JPEG decoded entirely on the host (MDC_GPU_JPEG=0):", flush=True)
        print(run("reader_rate_fast", d, 3, "batch", env={"MDC_GPU_JPEG": "0"}), flush=True)
        print("-- getImages, GPU JPEG stage 1 (host: Huffman decoding; device: dequantisation + inverse DCT; MDC_GPU_JPEG=1):", flush=True)
        print(run("reader_rate_fast", d, 3, "batch", env={"MDC_GPU_JPEG": "1"}), flush=True)
        print("-- getImages, GPU JPEG stage 2 (host: markers + byte stuffing only; device: Huffman decoding + inverse DCT; the default):", flush=True)
    print(run("reader_rate_fast", d, 3, "batch"), flush=True)
    print("-- getImagesDevice: the same pipeline, results LEFT IN HBM (nothing crosses PCIe on the way out):", flush=True)
    print(run("reader_rate_fast", d, 5, "device"), flush=True)
    print("-- getImagesDevice + box levels 1-3 + gradient images of every level (DSO hand-off), all left in HBM:", flush=True)
    print(run("reader_rate_fast", d, 5, "device_dso"), flush=True)
    for t in os.environ.get("MDC_RATE_THREADS", "").split(","):  # decode threads sweep of the device path (what binds it?)
        if t:
            print("-- getImagesDevice with %s decode threads:" % t, flush=True)
            print(run("reader_rate_fast", d, 5, "device", env={"MDC_READER_THREADS": t}), flush=True)
