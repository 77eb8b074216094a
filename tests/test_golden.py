"""Golden vectors generated from the reference itself (tests/golden/make_golden.py):
the C oracle (CPU, always) and the HIP path (GPU) must reproduce them bit for bit.
These are the pins that do not need /root/reference at test time."""
import glob
import itertools
import os

import numpy as np
import pytest

from conftest import bits_equal

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def u32(a):
    return np.asarray(a, np.float32).view(np.uint32)


def materialise(g, d):
    from mono_dataset_code_amd import synth

    synth.write_camera(os.path.join(d, "camera.txt"), tuple(str(x) for x in g["camera_lines"]))
    open(os.path.join(d, "pcalib.txt"), "w").write(str(g["pcalib_text"]))
    synth.write_png_gray(os.path.join(d, "vignette.png"), g["vignette_image"])


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path, oracle, tmp_path):
    g = np.load(path)
    d = str(tmp_path)
    materialise(g, d)
    W, H, w, h = (int(x) for x in g["dims"])
    cam = oracle.parse_camera(os.path.join(d, "camera.txt"))
    assert cam["valid"] and (cam["in_w"], cam["in_h"], cam["out_w"], cam["out_h"]) == (W, H, w, h)
    t = oracle.fov_setup(cam)
    assert np.array_equal(u32(t["remap_x"]), u32(g["remap_x"])) and np.array_equal(u32(t["remap_y"]), u32(g["remap_y"]))
    assert np.array_equal(u32(t["K_rect"]), u32(g["K_rect"])) and np.array_equal(u32(t["out_calib"]), u32(g["out_calib"]))
    ginv, gf = oracle.photo_gamma(oracle.parse_pcalib(os.path.join(d, "pcalib.txt")))
    assert np.array_equal(u32(ginv), u32(g["ginv"])) and np.array_equal(u32(gf), u32(g["g"]))
    vm, vi = oracle.photo_vignette(g["vignette_image"])
    assert np.array_equal(u32(vm), u32(g["vignette_map"])) and np.array_equal(u32(vi), u32(g["vignette_inv"]))
    x, y = g["distort_in"][0].copy(), g["distort_in"][1].copy()
    oracle.distort(cam, t["out_calib"], x, y)
    assert np.array_equal(u32(x), u32(g["distort_out"][0])) and np.array_equal(u32(y), u32(g["distort_out"][1]))
    for rect, gg, v, o in itertools.product((0, 1), repeat=4):
        for f, raw in enumerate(g["frames"]):
            got = oracle.get_image(raw, W, H, w, h, ginv, vi, True, True, t["remap_x"], t["remap_y"], rect, gg, v, o)
            assert bits_equal(got, g["out_%d%d%d%d" % (rect, gg, v, o)][f]), (rect, gg, v, o, f)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_host_classes_reproduce_golden_tables(path, tmp_path):
    """The product's host side (C++ classes) builds the reference's tables from the same files."""
    from mono_dataset_code_amd import capi

    g = np.load(path)
    d = str(tmp_path)
    materialise(g, d)
    W, H, w, h = (int(x) for x in g["dims"])
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), W, H)
    rx, ry = fov.remap()
    assert np.array_equal(u32(rx), u32(g["remap_x"])) and np.array_equal(u32(ry), u32(g["remap_y"]))
    assert np.array_equal(u32(fov.intrinsics()["K_rect"]), u32(g["K_rect"]))
    assert np.array_equal(u32(photo.ginv()), u32(g["ginv"])) and np.array_equal(u32(photo.g()), u32(g["g"]))
    assert np.array_equal(u32(photo.vignette()[1]), u32(g["vignette_inv"]))


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_path_reproduces_golden(path, tmp_path):
    """All 16 flag combinations through the C ABI (fused host call) against the reference's outputs."""
    from mono_dataset_code_amd import capi

    g = np.load(path)
    d = str(tmp_path)
    materialise(g, d)
    W, H, w, h = (int(x) for x in g["dims"])
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), W, H)
    assert fov.has_gpu() and photo.has_gpu()
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    for rect, gg, v, o in itertools.product((0, 1), repeat=4):
        flags = (capi.RECTIFY * rect) | (capi.GAMMA * gg) | (capi.VIGNETTE * v) | (capi.KILL_OVEREXPOSED * o)
        for f, raw in enumerate(g["frames"]):
            out = np.full(w * h if rect else W * H, -3.0, np.float32)
            ctx.process_host(np.ascontiguousarray(raw), out, flags)
            assert bits_equal(out, g["out_%d%d%d%d" % (rect, gg, v, o)][f]), (rect, gg, v, o, f)
    # and through the two class methods, as the reference's reader composes them
    for f, raw in enumerate(g["frames"]):
        tmp, out = np.zeros(W * H, np.float32), np.zeros(w * h, np.float32)
        photo.unmap(np.ascontiguousarray(raw), tmp, 1, 1, 1)
        fov.undistort(tmp, out)
        assert bits_equal(out, g["out_1111"][f])
