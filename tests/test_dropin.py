"""Drop-in check at the reference's own call site (BASELINE.json configs[0]).

The reference's UNMODIFIED src/BenchmarkDatasetReader.h is compiled by oracle/Makefile
under a headless playDataset-style driver (tests/dropin/playback_headless.cpp), once
against the reference's classes (playback_ref) and once against this repo's drop-in
classes + libmdc_host.so (playback_mdc).  On a synthetic sequence folder:
  CPU  : playback_ref (the reference's CPU path end to end) == the C oracle
  GPU  : playback_mdc (our classes, HIP kernels)            == playback_ref, byte for byte
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "playback_ref")
MDC_BIN = os.path.join(ROOT, "oracle", "_ref", "playback_mdc")
FAST_BIN = os.path.join(ROOT, "oracle", "_ref", "sequence_fast")
FLAGS = ["1111", "0111", "1000", "0000", "1110", "1010", "0001", "1100"]
N_FRAMES = 3
CAM = ("0.349153 0.436593 0.493140 0.499021 0.933271", "320 256", "crop", "192 144")


@pytest.fixture(scope="module")
def sequence(tmp_path_factory):
    from mono_dataset_code_amd import synth

    d = str(tmp_path_factory.mktemp("sequence_synth"))
    synth.write_sequence_calibration(d, CAM, vignette_bits=16, n_times=N_FRAMES)
    os.makedirs(os.path.join(d, "images"))
    frames = [synth.noise_frames(11, 1, 320 * 256)[0], synth.smooth_frame(320, 256, 0.7), synth.smooth_frame(320, 256, 2.1, blobs=False)]
    for i, f in enumerate(frames):
        synth.write_png_gray(os.path.join(d, "images", "%05d.png" % i), f.reshape(256, 320))
    return d, frames


def run_playback(binary, folder, out):
    r = subprocess.run([binary, folder, out] + FLAGS, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def parse(path):
    raw = open(path, "rb").read()
    pos, recs = 0, []
    while pos < len(raw):
        w, h, idx, a = np.frombuffer(raw, np.int32, 4, pos)
        ts = np.frombuffer(raw, np.float64, 1, pos + 16)[0]
        ex = np.frombuffer(raw, np.float32, 1, pos + 24)[0]
        img = np.frombuffer(raw, np.float32, w * h, pos + 28)
        recs.append((int(w), int(h), int(idx), int(a), float(ts), float(ex), img))
        pos += 28 + 4 * w * h
    return recs


def need(binary):
    if not os.path.exists(binary):
        pytest.skip("%s not built (needs /root/reference at build time)" % binary)


@pytest.fixture(scope="module")
def zipped_sequence(sequence, tmp_path_factory):
    """The same sequence with its frames in images.zip (deflated, archive order reversed) instead of images/."""
    import shutil
    import zipfile

    d, frames = sequence
    z = str(tmp_path_factory.mktemp("sequence_zipped"))
    for f in ("camera.txt", "pcalib.txt", "vignette.png", "times.txt"):
        shutil.copy(os.path.join(d, f), z)
    with zipfile.ZipFile(os.path.join(z, "images.zip"), "w", zipfile.ZIP_DEFLATED) as a:
        for n in sorted(os.listdir(os.path.join(d, "images")), reverse=True):
            a.write(os.path.join(d, "images", n), n)
    return z, frames


@pytest.mark.parametrize("zipped", [False, True])
def test_reference_reader_cpu_path_matches_oracle(sequence, zipped_sequence, oracle, tmp_path, zipped):
    need(REF_BIN)
    d, frames = zipped_sequence if zipped else sequence
    out = str(tmp_path / "ref.bin")
    log = run_playback(REF_BIN, d, out)
    assert "PLAYBACK 3 images, 320x256 -> 192x144" in log
    assert ("got 3 entries and 3 files from zipfile!" in log) == zipped
    recs = parse(out)
    assert len(recs) == len(FLAGS) * N_FRAMES
    cam = oracle.parse_camera(os.path.join(d, "camera.txt"))
    t = oracle.fov_setup(cam)
    ginv, _ = oracle.photo_gamma(oracle.parse_pcalib(os.path.join(d, "pcalib.txt")))
    from mono_dataset_code_amd import synth
    vinv = oracle.photo_vignette(synth.vignette_image(320, 256, 16))[1]
    for k, (w, h, idx, a, ts, ex, img) in enumerate(recs):
        fl = [int(c) for c in FLAGS[k // N_FRAMES]]
        assert idx == k % N_FRAMES and (w, h) == ((192, 144) if fl[0] else (320, 256))
        assert abs(ts - (1000.0 + idx / 20.0)) < 1e-9 and abs(ex - (1.0 + 0.01 * idx)) < 1e-6
        want = oracle.get_image(frames[idx], 320, 256, 192, 144, ginv, vinv, True, True, t["remap_x"], t["remap_y"], *fl)
        assert bits_equal(img, want), (FLAGS[k // N_FRAMES], idx)


@pytest.mark.gpu
def test_dropin_classes_under_unmodified_reader_match_reference(sequence, tmp_path):
    need(REF_BIN)
    need(MDC_BIN)
    d, _ = sequence
    a, b = str(tmp_path / "ref.bin"), str(tmp_path / "mdc.bin")
    log_ref = run_playback(REF_BIN, d, a)
    log_mdc = run_playback(MDC_BIN, d, b)
    assert "no GPU context" not in log_mdc, "the drop-in classes must run on the GPU"
    ra, rb = parse(a), parse(b)
    assert len(ra) == len(rb) == len(FLAGS) * N_FRAMES
    for x, y in zip(ra, rb):
        assert x[:6] == y[:6]
        assert bits_equal(x[6], y[6]), x[:4]
    # same chatter on stdout (the classes print the reference's messages)
    keep = lambda s: [l for l in s.splitlines() if l.startswith(("PLAYBACK", "Input resolution", "Output resolution", "Out:", "new K", "old K", "Successfully"))]
    assert keep(log_ref) == keep(log_mdc)


@pytest.mark.gpu
def test_fused_sequence_program_matches_reference_getimage(sequence, tmp_path):
    """INTEGRATION.md section B compiled as C++ (tests/dropin/sequence_fast.cpp): the unmodified reader decodes,
    mdc_bind_objects + one mdc_process_frames_host call per flag set produce every getImage() result of the
    reference, byte for byte, from page-locked buffers."""
    need(REF_BIN)
    need(FAST_BIN)
    d, _ = sequence
    a, b = str(tmp_path / "ref.bin"), str(tmp_path / "fast.bin")
    run_playback(REF_BIN, d, a)
    log = run_playback(FAST_BIN, d, b)
    assert "SEQUENCE_FAST 3 images, 320x256 -> 192x144" in log
    ra, rb = parse(a), parse(b)
    assert len(ra) == len(rb) == len(FLAGS) * N_FRAMES
    for x, y in zip(ra, rb):
        assert x[:6] == y[:6]
        assert bits_equal(x[6], y[6]), x[:4]
