import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_build():
    """Everything under test is native: build it (no-op when up to date)."""
    import __graft_entry__

    __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import loader

    return loader.Oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference's own code (oracle/_ref).  Present in the build container and,
    as a prebuilt .so, on the GPU box; skipped only if neither holds."""
    from oracle import loader

    try:
        return loader.Ref()
    except OSError as e:
        pytest.skip(str(e))


# ---- calibration fixtures -------------------------------------------------------------
# name -> (camera.txt lines, vignette bits)
CAMERAS = {
    # the bench configuration
    "full_1280_to_640": (("0.349153 0.436593 0.493140 0.499021 0.933271", "1280 1024", "0.4 0.53 0.5 0.5 0", "640 480"), 16),
    # small ones for exhaustive flag sweeps (input width multiple of 16 -> tiled kernel)
    "small_explicit": (("0.349153 0.436593 0.493140 0.499021 0.933271", "320 256", "0.4 0.53 0.5 0.5 0", "160 120"), 16),
    "small_crop": (("0.349153 0.436593 0.493140 0.499021 0.933271", "320 256", "crop", "200 152"), 8),
    "small_full_black": (("0.349153 0.436593 0.493140 0.499021 0.933271", "320 256", "full", "208 160"), 16),
    "small_pinhole": (("0.5 0.6 0.5 0.5 0", "320 256", "crop", "130 70"), 16),
    # ragged: input width not a multiple of 16 (gather kernel), odd output size
    "ragged": (("0.349153 0.436593 0.493140 0.499021 0.933271", "322 250", "0.4 0.53 0.5 0.5 0", "157 93"), 16),
    # output made of whole 64 x {16,32,64} tiles (fused pyramid path), black border pixels
    "pyr_whole_black": (("0.349153 0.436593 0.493140 0.499021 0.933271", "320 256", "full", "192 128"), 16),
    # BASELINE.json configs[4]: rectify 1280x1024 -> 1280x1024, the base of the 4-level pyramid (bench.py --workload pyramid)
    "full_1280_to_1280": (("0.349153 0.436593 0.493140 0.499021 0.933271", "1280 1024", "0.4 0.53 0.5 0.5 0", "1280 1024"), 16),
    # full-size cameras beyond the bench's one: `crop`, `full` (black border outputs: the BLACK instantiations of the
    # 1024-thread tiles), a pinhole (omega = 0), the wide lens of the public dataset's other camera (recalled, not in /root/reference:
    # a representative synthetic line), and output sizes whose last tile column is ragged (752 = 5 x 128 + 112, 1000 = 7 x 128 + 104)
    "full_1280_crop": (("0.349153 0.436593 0.493140 0.499021 0.933271", "1280 1024", "crop", "640 480"), 16),
    "full_1280_full_black": (("0.349153 0.436593 0.493140 0.499021 0.933271", "1280 1024", "full", "640 480"), 16),
    "full_1280_pinhole": (("0.5 0.6 0.5 0.5 0", "1280 1024", "crop", "640 480"), 8),
    "full_1280_wide": (("0.535719 0.669567 0.493249 0.500409 0.897966", "1280 1024", "crop", "640 480"), 16),
    "full_1280_to_752": (("0.349153 0.436593 0.493140 0.499021 0.933271", "1280 1024", "0.4 0.53 0.5 0.5 0", "752 480"), 16),
    "full_1280_to_1000": (("0.535719 0.669567 0.493249 0.500409 0.897966", "1280 1024", "full", "1000 700"), 16),
    # magnifying remap (output larger than input)
    "upsample": (("0.349153 0.436593 0.493140 0.499021 0.5", "160 128", "crop", "320 256"), 16),
    # two-stage kernel cases: a magnifying 'full' remap made of whole 128x16 / 64x32 tiles with a wide black border (0.45-0.49
    # staged source pixels per output: the automatic choice), and a ~1:1 crop whose windows need 4 convert passes per wave
    "mag_full_black": (("0.349153 0.436593 0.493140 0.499021 0.933271", "320 256", "full", "640 512"), 16),
    "crop_384": (("0.349153 0.436593 0.493140 0.499021 0.933271", "320 256", "crop", "384 256"), 8),
    # strip kernel cases (windows of a 128 x 8 output tile <= 128 chunks): 4x magnifying 'full' remaps with a third of
    # the outputs black, whole 128 x 8 tiles (640 x 512) and a ragged last tile column (520 wide), two DMA rounds (67 / 80 chunks)
    "mag4_full_black": (("0.349153 0.436593 0.493140 0.499021 0.933271", "160 128", "full", "640 512"), 16),
    "mag4_ragged": (("0.349153 0.436593 0.493140 0.499021 0.933271", "160 128", "full", "520 384"), 8),
}


@pytest.fixture(scope="session")
def calib_dirs(tmp_path_factory):
    from mono_dataset_code_amd import synth

    out = {}
    for name, (lines, bits) in CAMERAS.items():
        d = str(tmp_path_factory.mktemp(name))
        synth.write_sequence_calibration(d, lines, vignette_bits=bits)
        out[name] = d
    return out


def bits_equal(a, b):
    """Bitwise equality of float arrays, NaNs compared as a mask (payload/sign of a
    propagated NaN is not specified by either side)."""
    a = np.asarray(a, np.float32).ravel()
    b = np.asarray(b, np.float32).ravel()
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32)))


def test_frames(w, h, n_noise=2):
    """noise frames + a smooth frame with saturated blobs + all-255 + all-0."""
    from mono_dataset_code_amd import synth

    fr = [f for f in synth.noise_frames(3, n_noise, w * h)]
    fr.append(synth.smooth_frame(w, h, 0.3))
    fr.append(np.full(w * h, 255, np.uint8))
    fr.append(np.zeros(w * h, np.uint8))
    return fr
