"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

The oracle is oracle/mdc_oracle.c (plain-C restatement of the reference, itself
pinned bit-for-bit against the reference build in test_oracle_vs_ref.py).
Tolerance: none -- values must be bit-identical, the NaN (overexposure) mask and
the black-pixel zeros exact.  (BASELINE.json asks for <= 1e-4 relative.)
"""
import itertools
import os

import numpy as np
import pytest

from conftest import CAMERAS, bits_equal, test_frames as make_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


class Setup:
    def __init__(self, name, calib_dirs, oracle):
        from mono_dataset_code_amd import capi

        d = calib_dirs[name]
        self.cam = oracle.parse_camera(os.path.join(d, "camera.txt"))
        assert self.cam["valid"]
        self.W, self.H, self.w, self.h = self.cam["in_w"], self.cam["in_h"], self.cam["out_w"], self.cam["out_h"]
        self.fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
        self.photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), self.W, self.H)
        assert self.fov.is_valid() and self.photo.valid() == 3
        assert self.fov.has_gpu() and self.photo.has_gpu(), "native GPU context missing"
        self.rx, self.ry = self.fov.remap()
        self.ginv = self.photo.ginv()
        self.vinv = self.photo.vignette()[1]
        self.ctx = capi.Context(0)
        self.ctx.bind(self.fov, self.photo)

    def want(self, oracle, raw, rect, g, v, o):
        return oracle.get_image(raw, self.W, self.H, self.w, self.h, self.ginv, self.vinv, True, True, self.rx, self.ry,
                                rect, g, v, o)


@pytest.fixture(scope="module")
def setups(calib_dirs, oracle):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Setup(name, calib_dirs, oracle)
        return cache[name]

    return get


SMALL = [n for n in CAMERAS if not n.startswith("full_1280")]


@pytest.mark.parametrize("name", SMALL)
def test_process_host_all_flags(name, setups, oracle):
    """All 16 combinations of getImage's four bools x 5 frame kinds, fused host call."""
    from mono_dataset_code_amd import capi

    s = setups(name)
    for raw in make_frames(s.W, s.H):
        for rect, g, v, o in itertools.product((0, 1), repeat=4):
            flags = (capi.RECTIFY * rect) | (capi.GAMMA * g) | (capi.VIGNETTE * v) | (capi.KILL_OVEREXPOSED * o)
            out = np.full(s.w * s.h if rect else s.W * s.H, -7.0, np.float32)
            s.ctx.process_host(raw, out, flags)
            assert bits_equal(out, s.want(oracle, raw, rect, g, v, o)), (name, rect, g, v, o)


@pytest.mark.parametrize("name", SMALL)
def test_tiled_and_gather_kernels_agree(name, setups, oracle, torch_cuda):
    """Device batch API, both kernels, ragged batch sizes (1, 3, 17 frames)."""
    from mono_dataset_code_amd import capi

    torch = torch_cuda
    s = setups(name)
    info = s.ctx.info()
    frames = np.stack(make_frames(s.W, s.H, n_noise=14))  # 17 frames
    assert len(frames) == 17
    want = np.stack([s.want(oracle, f, 1, 1, 1, 1) for f in frames])
    d_in = torch.from_numpy(frames).cuda()
    flags = capi.RECTIFY | capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED
    s.ctx.set_option(capi.OPT_TILE_ROWS, 16)
    info = s.ctx.info()
    kernels = [capi.KERNEL_GATHER] + ([capi.KERNEL_TILED] if info.tiled else [])
    if name not in ("ragged", "small_full_black", "pyr_whole_black"):  # width % 16 != 0 / windows too large for 64x16 tiles
        assert info.tiled, "tiled kernel should be plannable for %s" % name
    for k in kernels:
        s.ctx.set_option(capi.OPT_KERNEL, k)
        for rows, order, nbuf, cols in (((32, capi.ORDER_BANDS, 0, 64), (16, capi.ORDER_ROWS, 4, 64), (32, capi.ORDER_IDENTITY, 3, 64), (60, capi.ORDER_BANDS, 0, 64),
                                   (64, capi.ORDER_ROWS, 2, 64), (16, capi.ORDER_BANDS, 2, 64), (32, capi.ORDER_ROWS, 4, 64), (60, capi.ORDER_ROWS, 2, 64),
                                   (32, capi.ORDER_BLOCKS2D, 0, 128), (16, capi.ORDER_BANDS, 4, 128), (32, capi.ORDER_ROWS, 3, 128), (16, capi.ORDER_BLOCKS2D, 2, 128),
                                   (32, capi.ORDER_BLOCKS2D, 2, 64), (32, capi.ORDER_BANDS, 2, 128), (60, capi.ORDER_BANDS, 3, 64))
                                  if k == capi.KERNEL_TILED else ((32, capi.ORDER_BANDS, 0, 64),)):
          s.ctx.set_option(capi.OPT_TILE_COLS, cols)
          s.ctx.set_option(capi.OPT_TILE_ROWS, rows)
          s.ctx.set_option(capi.OPT_TILE_ORDER, order)
          s.ctx.set_option(capi.OPT_WINDOW_BUFFERS, nbuf)
          if k == capi.KERNEL_TILED and not s.ctx.info().tiled:
              continue
          for n, fpb, il in ((1, 0, 0), (3, 2, 1), (17, 0, 0), (17, 5, 1), (17, 5, 0), (17, 0, 1)):
            s.ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, fpb)
            s.ctx.set_option(capi.OPT_FRAME_INTERLEAVE, il)
            d_out = torch.full((n, s.w * s.h), -7.0, dtype=torch.float32, device="cuda")
            s.ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), n, flags, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert bits_equal(d_out.cpu().numpy(), want[:n]), (name, k, rows, cols, order, nbuf, n, fpb, il)
    s.ctx.set_option(capi.OPT_TILE_COLS, 0)
    s.ctx.set_option(capi.OPT_TILE_ORDER, capi.ORDER_BANDS)
    s.ctx.set_option(capi.OPT_WINDOW_BUFFERS, 0)
    s.ctx.set_option(capi.OPT_FRAME_INTERLEAVE, 0)
    s.ctx.set_option(capi.OPT_TILE_ROWS, 0)
    s.ctx.set_option(capi.OPT_KERNEL, capi.KERNEL_AUTO)
    s.ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, 0)


@pytest.mark.parametrize("name", ["small_explicit", "ragged"])
def test_unmap_and_undistort_host(name, setups, oracle):
    """The two reference methods separately, through the C ABI host calls."""
    from mono_dataset_code_amd import capi

    s = setups(name)
    for raw in make_frames(s.W, s.H):
        for g, v, o in itertools.product((0, 1), repeat=3):
            out = np.full(s.W * s.H, -7.0, np.float32)
            s.ctx.unmap_host(raw, out, (capi.GAMMA * g) | (capi.VIGNETTE * v) | (capi.KILL_OVEREXPOSED * o))
            want = oracle.unmap(raw, s.ginv, s.vinv, True, True, g, v, o)
            assert bits_equal(out, want), (name, g, v, o)
        # undistort<unsigned char> and undistort<float> (float input = the unmapped frame incl. NaN/inf)
        out = np.full(s.w * s.h, -7.0, np.float32)
        s.ctx.undistort_host(raw, out)
        assert bits_equal(out, oracle.undistort(raw, s.rx, s.ry, s.W))
        fin = oracle.unmap(raw, s.ginv, s.vinv, True, True, 1, 1, 1)
        s.ctx.undistort_host(fin, out)
        assert bits_equal(out, oracle.undistort(fin, s.rx, s.ry, s.W))


@pytest.mark.parametrize("name", ["small_explicit", "small_full_black", "upsample", "ragged"])
def test_undistort_f32_batch_tiled_and_gather(name, setups, oracle, torch_cuda):
    """UndistorterFOV::undistort<float> on device batches: the LDS-tiled float kernel (every tile shape and
    buffer count) and the gather kernel against the oracle, inputs with NaN / inf (unmapped frames)."""
    from mono_dataset_code_amd import capi

    torch = torch_cuda
    s = setups(name)
    raws = make_frames(s.W, s.H, n_noise=4)
    fins = np.stack([oracle.unmap(r, s.ginv, s.vinv, True, True, 1, 1, 1) for r in raws])
    want = np.stack([oracle.undistort(f, s.rx, s.ry, s.W) for f in fins])
    n = len(fins)
    d_in = torch.from_numpy(fins).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for kernel, rows, nbuf, fpb in ((capi.KERNEL_GATHER, 32, 0, 0), (capi.KERNEL_AUTO, 32, 0, 0), (capi.KERNEL_AUTO, 16, 3, 2),
                                    (capi.KERNEL_AUTO, 60, 2, 0), (capi.KERNEL_AUTO, 64, 0, 3), (capi.KERNEL_AUTO, 32, 4, 1)):
        s.ctx.set_option(capi.OPT_KERNEL, kernel)
        s.ctx.set_option(capi.OPT_TILE_ROWS, rows)
        s.ctx.set_option(capi.OPT_WINDOW_BUFFERS, nbuf)
        s.ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, fpb)
        d_out = torch.full((n, s.w * s.h), -7.0, dtype=torch.float32, device="cuda")
        s.ctx.undistort_batch_f32(d_in.data_ptr(), d_out.data_ptr(), n, st)
        torch.cuda.synchronize()
        assert bits_equal(d_out.cpu().numpy(), want), (name, kernel, rows, nbuf, fpb)
    s.ctx.set_option(capi.OPT_KERNEL, capi.KERNEL_AUTO)
    s.ctx.set_option(capi.OPT_TILE_ROWS, 0)
    s.ctx.set_option(capi.OPT_WINDOW_BUFFERS, 0)
    s.ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, 0)


def test_cxx_classes_match_reference(calib_dirs, ref, oracle):
    """The drop-in C++ classes against the reference's own classes, method by method."""
    from mono_dataset_code_amd import capi

    d = calib_dirs["small_full_black"]
    cam = os.path.join(d, "camera.txt")
    ours, theirs = capi.UndistorterFOV(cam), ref.fov(cam)
    W, H, w, h = theirs.dims()
    po = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), W, H)
    pt = ref.photo(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), W, H)
    assert ours.has_gpu() and po.has_gpu()
    for raw in make_frames(W, H):
        for g, v, o in itertools.product((0, 1), repeat=3):
            a, b = np.zeros(W * H, np.float32), np.zeros(W * H, np.float32)
            po.unmap(raw, a, g, v, o)
            pt.unmap(raw.copy(), b, g, v, o)
            assert bits_equal(a, b), (g, v, o)
            a2, b2 = np.zeros(w * h, np.float32), np.zeros(w * h, np.float32)
            ours.undistort(a, a2)
            theirs.undistort(b, b2)
            assert bits_equal(a2, b2), (g, v, o)
        a2, b2 = np.zeros(w * h, np.float32), np.zeros(w * h, np.float32)
        ours.undistort(raw, a2)
        theirs.undistort(raw, b2)
        assert bits_equal(a2, b2)
    # wrong sizes: output untouched, like the reference (FOVUndistorter.cpp:327-338)
    a2 = np.full(w * h, 5.0, np.float32)
    ours.undistort(np.zeros(W * H - 1, np.uint8), a2)
    assert np.all(a2 == 5.0)


@pytest.mark.parametrize("name", ["full_1280_to_640", "full_1280_to_1280"])
def test_full_size_flag_matrix_against_the_reference_build(name, calib_dirs, ref, torch_cuda):
    """All 16 getImage switch combinations (src/BenchmarkDatasetReader.h:207-241) x {noise, smooth + saturated blobs, all-255, all-0}
    at 1280x1024 -> 640x480 and -> 1280x1024, through mdc_process_batch_device AND through the class methods, compared straight with
    the REFERENCE BUILD (oracle/_ref/libmdc_ref.so: its own classes, its own tables from the same calibration files) -- not with
    the C restatement fed with our tables."""
    from mono_dataset_code_amd import capi

    torch = torch_cuda
    d = calib_dirs[name]
    cam, pc, vg = os.path.join(d, "camera.txt"), os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png")
    rfov = ref.fov(cam)
    W, H, w, h = rfov.dims()
    assert (W, H) == (1280, 1024)
    rphoto = ref.photo(pc, vg, W, H)
    fov, photo = capi.UndistorterFOV(cam), capi.PhotometricUndistorter(pc, vg, W, H)
    assert fov.has_gpu() and photo.has_gpu()
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    frames = np.stack(make_frames(W, H, n_noise=1))  # noise, smooth + blobs, all-255, all-0
    n = len(frames)
    assert n == 4
    d_in = torch.from_numpy(frames).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for rect, g, v, o in itertools.product((0, 1), repeat=4):
        flags = (capi.RECTIFY * rect) | (capi.GAMMA * g) | (capi.VIGNETTE * v) | (capi.KILL_OVEREXPOSED * o)
        nout = w * h if rect else W * H
        d_out = torch.full((n, nout), -7.0, dtype=torch.float32, device="cuda")
        ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), n, flags, st)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        for f in range(n):
            want = ref.get_image(rfov, rphoto, frames[f].copy(), rect, g, v, o)
            assert bits_equal(got[f], want), (name, "batch", f, rect, g, v, o)
    # the class methods, composed as the reference's reader composes them (:212-233)
    for f in range(n):
        for g, v, o in itertools.product((0, 1), repeat=3):
            a, b = np.zeros(W * H, np.float32), np.zeros(W * H, np.float32)
            photo.unmap(frames[f], a, g, v, o)
            rphoto.unmap(frames[f].copy(), b, g, v, o)
            assert bits_equal(a, b), (name, "unMapImage", f, g, v, o)
            if (g, v, o) in ((1, 1, 1), (0, 0, 1), (1, 0, 0)):
                a2, b2 = np.zeros(w * h, np.float32), np.zeros(w * h, np.float32)
                fov.undistort(a, a2)
                rfov.undistort(b, b2)
                assert bits_equal(a2, b2), (name, "undistort<float>", f, g, v, o)
        a2, b2 = np.zeros(w * h, np.float32), np.zeros(w * h, np.float32)
        fov.undistort(frames[f], a2)
        rfov.undistort(frames[f], b2)
        assert bits_equal(a2, b2), (name, "undistort<unsigned char>", f)
    ctx.close()


FULL_SIZE_CAMERAS = ["full_1280_crop", "full_1280_full_black", "full_1280_pinhole", "full_1280_wide", "full_1280_to_752", "full_1280_to_1000"]


@pytest.mark.parametrize("name", FULL_SIZE_CAMERAS)
def test_full_size_other_cameras_against_the_reference_build(name, calib_dirs, ref, torch_cuda):
    """1280x1024 frames through cameras other than the bench's: `crop`, `full` (black outputs), a pinhole, a wide lens, ragged last tile
    columns -- the output-K selection of src/FOVUndistorter.cpp:144-218 and the border rules of :235-251 at full size.  All 16 getImage
    switch combinations x {noise, smooth + blobs, all-255, all-0} on the library's own plan, then the rectifying combinations again on the
    128 x 32 and 128 x 16 tile plans set explicitly and on the plan mdc_tune_device picks, each against the REFERENCE BUILD bit for bit."""
    from mono_dataset_code_amd import capi, synth

    torch = torch_cuda
    d = calib_dirs[name]
    cam, pc, vg = os.path.join(d, "camera.txt"), os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png")
    rfov = ref.fov(cam)
    W, H, w, h = rfov.dims()
    assert (W, H) == (1280, 1024) and rfov.is_valid()
    rphoto = ref.photo(pc, vg, W, H)
    fov, photo = capi.UndistorterFOV(cam), capi.PhotometricUndistorter(pc, vg, W, H)
    assert fov.has_gpu() and photo.has_gpu() and fov.dims() == (W, H, w, h)
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    info = ctx.info()
    if name == "full_1280_full_black":
        assert info.n_black > 0  # the point of this camera
    frames = np.stack(make_frames(W, H, n_noise=1))
    n = len(frames)
    d_in = torch.from_numpy(frames).cuda()
    st = torch.cuda.current_stream().cuda_stream
    wants = {}

    def want(f, rect, g, v, o):
        key = (f, rect, g, v, o)
        if key not in wants:
            wants[key] = ref.get_image(rfov, rphoto, frames[f].copy(), rect, g, v, o)
        return wants[key]

    def check(combos, tag):
        for rect, g, v, o in combos:
            flags = (capi.RECTIFY * rect) | (capi.GAMMA * g) | (capi.VIGNETTE * v) | (capi.KILL_OVEREXPOSED * o)
            nout = w * h if rect else W * H
            d_out = torch.full((n, nout), -7.0, dtype=torch.float32, device="cuda")
            ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), n, flags, st)
            torch.cuda.synchronize()
            got = d_out.cpu().numpy()
            for f in range(n):
                assert bits_equal(got[f], want(f, rect, g, v, o)), (name, tag, f, rect, g, v, o)

    check(list(itertools.product((0, 1), repeat=4)), "library plan %dx%d tiled=%d strip=%d" % (info.tile_w, info.tile_h, info.tiled, info.two_stage))
    rectifying = [(1, 1, 1, 1), (1, 0, 0, 0), (1, 1, 0, 0), (1, 0, 0, 1)]
    ran = []
    for cols, rows in ((128, 32), (128, 16), (64, 32)):
        ctx.set_option(capi.OPT_TILE_COLS, cols)
        ctx.set_option(capi.OPT_TILE_ROWS, rows)
        i2 = ctx.info()
        if i2.tiled and (i2.tile_w, i2.tile_h) == (cols, rows) and not i2.two_stage:
            ran.append((cols, rows))
            check(rectifying, "plan %dx%d" % (cols, rows))
    if name == "full_1280_to_752":  # the bench camera's K at another output size: both headline shapes plan (`crop` / `full` at 640 x 480
        assert (128, 32) in ran and (128, 16) in ran, ran  # shrink 5.8 source pixels into one output: windows beyond any tile's LDS budget)
    print("PLANS %s: library %dx%d tiled=%d strip=%d black=%d; explicit shapes that planned: %s" % (name, info.tile_w, info.tile_h, info.tiled, info.two_stage, info.n_black, ran))
    ctx.set_option(capi.OPT_TILE_COLS, 0)
    ctx.set_option(capi.OPT_TILE_ROWS, 0)
    if ctx.info().tiled and not ctx.info().two_stage:
        nb = 192  # the tuner wants a batch: noise frames, results checked on the first four (which are `frames`)
        big = torch.empty(nb * W * H, dtype=torch.uint8, device="cuda")
        big[:n * W * H] = d_in.view(-1)
        ctx.synth_frames(big.data_ptr() + n * W * H, 100, nb - n, W * H, synth.SEED, st)
        out = torch.empty(nb * w * h, dtype=torch.float32, device="cuda")
        t = ctx.tune(big.data_ptr(), out.data_ptr(), nb, capi.RECTIFY | capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED, st)
        assert t.candidates >= 1
        ctx.process_batch(big.data_ptr(), out.data_ptr(), nb, capi.RECTIFY | capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED, st)
        torch.cuda.synchronize()
        got = out.view(nb, w * h)[:n].cpu().numpy()
        for f in range(n):
            assert bits_equal(got[f], want(f, 1, 1, 1, 1)), (name, "tuned plan %dx%d fpb %d" % (t.tile_w, t.tile_h, t.frames_per_block), f)
        del big, out
        check(rectifying, "tuned plan %dx%d" % (t.tile_w, t.tile_h))
    ctx.close()


def test_full_size_config_and_properties(setups, oracle, torch_cuda):
    """BASELINE.json configs[1]/[2] at full size: oracle on 3 frames, then
    size-independent properties on a 64-frame batch."""
    from mono_dataset_code_amd import capi, synth

    torch = torch_cuda
    s = setups("full_1280_to_640")
    info = s.ctx.info()
    assert info.tiled and info.in_w == 1280 and info.out_w == 640
    n = 64
    npix, nout = s.W * s.H, s.w * s.h
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.empty(n * npix, dtype=torch.uint8, device="cuda")
    s.ctx.synth_frames(d_in.data_ptr(), 100, n, npix, synth.SEED, st)
    flags = capi.RECTIFY | capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED
    d_out = torch.empty(n * nout, dtype=torch.float32, device="cuda")
    s.ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), n, flags, st)
    torch.cuda.synchronize()
    frames = d_in.view(n, npix).cpu().numpy()
    assert np.array_equal(frames[:2], synth.noise_frames(100, 2, npix))  # HIP generator == numpy generator
    got = d_out.view(n, nout).cpu().numpy()
    for f in (0, 31, 63):
        assert bits_equal(got[f], s.want(oracle, frames[f], 1, 1, 1, 1)), f
    # property 1: batching is invisible -- frame-at-a-time launches give the same bytes
    d_one = torch.empty(nout, dtype=torch.float32, device="cuda")
    for f in (5, 40):
        s.ctx.process_batch(d_in.data_ptr() + f * npix, d_one.data_ptr(), 1, flags, st)
        torch.cuda.synchronize()
        assert bits_equal(d_one.cpu().numpy(), got[f])
    # property 2: gather and tiled kernels agree on the whole batch
    s.ctx.set_option(capi.OPT_KERNEL, capi.KERNEL_GATHER)
    d_g = torch.empty(n * nout, dtype=torch.float32, device="cuda")
    s.ctx.process_batch(d_in.data_ptr(), d_g.data_ptr(), n, flags, st)
    torch.cuda.synchronize()
    s.ctx.set_option(capi.OPT_KERNEL, capi.KERNEL_AUTO)
    assert bits_equal(d_g.cpu().numpy(), got)
    # property 3: NaN rate of uniform noise = 1-(255/256)^4 of valid outputs
    rate = np.isnan(got).mean()
    assert abs(rate - (1 - (255 / 256) ** 4)) < 2e-4
    # property 4: fused == unmap followed by undistort<float> (the reference's two-pass composition)
    d_tmp = torch.empty(2 * npix, dtype=torch.float32, device="cuda")
    d_two = torch.empty(2 * nout, dtype=torch.float32, device="cuda")
    s.ctx.unmap_batch(d_in.data_ptr(), d_tmp.data_ptr(), 2, flags, st)
    s.ctx.undistort_batch_f32(d_tmp.data_ptr(), d_two.data_ptr(), 2, st)
    torch.cuda.synchronize()
    assert bits_equal(d_two.cpu().numpy(), got[:2])
    # config 2: unMapImage only, full size
    want = oracle.unmap(frames[0], s.ginv, s.vinv, True, True, 1, 1, 1)
    assert bits_equal(d_tmp[:npix].cpu().numpy(), want)


def test_automatic_tile_shape_keeps_strong_distortion_on_the_tiled_kernel(setups):
    """'full' cameras have source windows too wide for the small tiles; the automatic choice walks down the candidate
    list to a shape whose windows fit instead of dropping to the gather kernel."""
    from mono_dataset_code_amd import capi

    legal = {(64, 16), (64, 32), (64, 60), (64, 64), (128, 16), (128, 32)}
    for name, fits_64x32 in (("small_explicit", True), ("small_full_black", False), ("pyr_whole_black", False)):
        s = setups(name)
        s.ctx.set_option(capi.OPT_TILE_COLS, 0)
        s.ctx.set_option(capi.OPT_TILE_ROWS, 0)
        info = s.ctx.info()
        assert info.tiled and (info.tile_w, info.tile_h) in legal, (name, info.tiled, info.tile_w, info.tile_h)
        assert info.window_buffers in (2, 3, 4) and info.lds_bytes <= 160 * 1024
        if fits_64x32:
            assert (info.tile_w, info.tile_h) == (128, 16)  # the first candidate
        s.ctx.set_option(capi.OPT_TILE_COLS, 64)
        s.ctx.set_option(capi.OPT_TILE_ROWS, 32)
        # (since the plan stages the exact chunk SET of every source row instead of one run per row, the bowed windows of
        # the 'full' cameras may fit the smaller tile after all: only "a camera known to fit must fit" is asserted)
        assert s.ctx.info().tiled or not fits_64x32
        s.ctx.set_option(capi.OPT_TILE_COLS, 0)  # rows forced, columns free: 128 x 32 takes the wide windows
        assert s.ctx.info().tiled and s.ctx.info().tile_h == 32
        s.ctx.set_option(capi.OPT_TILE_ROWS, 0)
    # undistort<float> has its own plan and shape
    i = setups("small_explicit").ctx.info()
    assert i.f32_tiled and (i.f32_tile_w, i.f32_tile_h) == (128, 16)


def test_tune_keeps_results_and_picks_a_plan(setups, oracle, torch_cuda):
    """mdc_tune_device: whatever plan wins, the outputs it leaves and all later launches are the oracle's bits."""
    from mono_dataset_code_amd import capi, synth

    torch = torch_cuda
    s = setups("full_1280_to_640")
    n, npix, nout = 96, s.W * s.H, s.w * s.h
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.empty(n * npix, dtype=torch.uint8, device="cuda")
    s.ctx.synth_frames(d_in.data_ptr(), 500, n, npix, synth.SEED, st)
    d_out = torch.full((n, nout), -7.0, dtype=torch.float32, device="cuda")
    r = s.ctx.tune(d_in.data_ptr(), d_out.data_ptr(), n, 15, st)
    assert r.candidates == 12 and (r.tile_w, r.tile_h) in ((128, 16), (64, 32), (128, 32)) and r.frames_per_block in (32, 64, 96, 128) and r.ms > 0
    info = s.ctx.info()
    assert info.tiled and (info.tile_w, info.tile_h) == (r.tile_w, r.tile_h)
    frames = d_in.view(n, npix).cpu().numpy()
    want = {f: s.want(oracle, frames[f], 1, 1, 1, 1) for f in (0, 57, n - 1)}
    for f, w in want.items():
        assert bits_equal(d_out[f].cpu().numpy(), w), ("left by the tuner", f)
    d_out.fill_(-7.0)
    s.ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), n, 15, st)
    torch.cuda.synchronize()
    for f, w in want.items():
        assert bits_equal(d_out[f].cpu().numpy(), w), ("after tuning", f)
    with pytest.raises(capi.MdcError):
        s.ctx.tune(d_in.data_ptr(), d_out.data_ptr(), n, 7, st)  # no MDC_RECTIFY
    for opt in (capi.OPT_TILE_COLS, capi.OPT_TILE_ROWS, capi.OPT_FRAMES_PER_BLOCK):
        s.ctx.set_option(opt, 0)


def test_tapered_tail_of_large_launches(setups, oracle, torch_cuda):
    """MDC_OPT_TAIL_TAPER: a launch of many rounds of workgroups ends on frame groups of 1/2, 1/4 and 1/8 of the frames per
    workgroup.  Frame counts that leave 0, 1 and fpb/8 + 3 frames for the last level, with 16 and 24 frames per workgroup: every
    frame of the batch equals the launch without the taper bit for bit (no frame skipped, none written twice with another
    group's data), and sampled frames equal the oracle."""
    from mono_dataset_code_amd import capi, synth

    torch = torch_cuda
    s = setups("full_1280_to_640")
    npix, nout = s.W * s.H, s.w * s.h
    st = torch.cuda.current_stream().cuda_stream
    nmax = 700
    d_in = torch.empty(nmax * npix, dtype=torch.uint8, device="cuda")
    s.ctx.synth_frames(d_in.data_ptr(), 7, nmax, npix, synth.SEED, st)
    d_a = torch.empty(nmax * nout, dtype=torch.float32, device="cuda")
    d_b = torch.empty(nmax * nout, dtype=torch.float32, device="cuda")
    try:
        for fpb, n in ((16, 512), (16, 513), (16, 517), (24, 700), (16, 300)):
            s.ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, fpb)
            for taper, dst in ((2, d_a), (1, d_b)):
                s.ctx.set_option(capi.OPT_TAIL_TAPER, taper)
                dst.fill_(-7.0)
                s.ctx.process_batch(d_in.data_ptr(), dst.data_ptr(), n, 15, st)
            torch.cuda.synchronize()
            assert torch.equal(d_a[: n * nout].view(torch.int32), d_b[: n * nout].view(torch.int32)), (fpb, n)
            assert bool((d_b[n * nout:] == -7.0).all()), (fpb, n, "wrote past the batch")
            frames = d_in.view(nmax, npix)
            for f in (0, n // 2, n - 9, n - 2, n - 1):
                assert bits_equal(d_b.view(nmax, nout)[f].cpu().numpy(), s.want(oracle, frames[f].cpu().numpy(), 1, 1, 1, 1)), (fpb, n, f)
    finally:
        s.ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, 0)
        s.ctx.set_option(capi.OPT_TAIL_TAPER, 0)


def test_batch_beyond_4gib(setups, oracle, torch_cuda):
    """A batch whose frames lie beyond 4 GiB from the base pointers (3400 frames: 4.46 GB in, 4.18 GB out):
    the per-frame buffer descriptors take a 48-bit base, lane offsets stay 32-bit.  Checked against the
    oracle on both sides of the 4 GiB line, for the fused and the unMapImage kernels."""
    from mono_dataset_code_amd import capi, synth

    torch = torch_cuda
    s = setups("full_1280_to_640")
    n = 3400
    npix, nout = s.W * s.H, s.w * s.h
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.empty(n * npix, dtype=torch.uint8, device="cuda")
    s.ctx.synth_frames(d_in.data_ptr(), 7, n, npix, synth.SEED, st)
    flags = capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED
    d_out = torch.zeros(n * nout, dtype=torch.float32, device="cuda")
    s.ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), n, flags | capi.RECTIFY, st)
    torch.cuda.synchronize()
    line_in, line_out = (1 << 32) // npix, (1 << 32) // (nout * 4)  # first frame past 4 GiB: input 3276, output 3495 (> n)
    picks = (0, line_in - 1, line_in, line_in + 1, n - 1)
    for f in picks:
        raw = synth.noise_frames(7 + f, 1, npix)[0]
        assert np.array_equal(d_in[f * npix:(f + 1) * npix].cpu().numpy(), raw), f
        assert bits_equal(d_out[f * nout:(f + 1) * nout].cpu().numpy(), s.want(oracle, raw, 1, 1, 1, 1)), f
    del d_out
    # unMapImage: 5.24 MB of output per frame -> the output crosses 4 GiB at frame 819
    m = 900
    d_un = torch.zeros(m * npix, dtype=torch.float32, device="cuda")
    s.ctx.process_batch(d_in.data_ptr(), d_un.data_ptr(), m, flags, st)
    torch.cuda.synchronize()
    for f in (0, 818, 819, 820, m - 1):
        raw = synth.noise_frames(7 + f, 1, npix)[0]
        assert bits_equal(d_un[f * npix:(f + 1) * npix].cpu().numpy(), oracle.unmap(raw, s.ginv, s.vinv, True, True, 1, 1, 1)), f


def test_batch_launch_is_graph_capturable(setups, oracle, torch_cuda):
    """mdc_process_batch_device does no allocation / synchronisation: it can be captured into a HIP graph on
    the caller's stream and replayed (a reader that re-uses its buffers pays one graph launch per batch)."""
    from mono_dataset_code_amd import capi

    torch = torch_cuda
    s = setups("small_crop")
    frames = np.stack(make_frames(s.W, s.H, n_noise=3))
    n = len(frames)
    flags = capi.RECTIFY | capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros((n, s.w * s.h), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        s.ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), n, flags, st.cuda_stream)  # warm (module load)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            s.ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), n, flags, torch.cuda.current_stream().cuda_stream)
        for rep in range(3):
            d_in.copy_(torch.from_numpy(np.roll(frames, rep, axis=0)))
            d_out.fill_(-7.0)
            g.replay()
            st.synchronize()
            got = d_out.cpu().numpy()
            for i in range(n):
                assert bits_equal(got[i], s.want(oracle, np.roll(frames, rep, axis=0)[i], 1, 1, 1, 1)), (rep, i)


@pytest.mark.parametrize("name", ["small_explicit", "small_crop", "small_full_black", "small_pinhole", "upsample", "full_1280_to_640"])
def test_distort_points_on_gpu_equals_host_libm(name, setups, oracle, calib_dirs, torch_cuda):
    """distortCoordinates on the GPU (fdlibm atanf restated in the kernel) against the oracle and against the
    class method (both on the host libm): bit for bit on 2*10^6 points that sweep every branch of atanf
    (inside the image, far outside, tiny radii, the exact centre)."""
    from mono_dataset_code_amd import capi

    s = setups(name)
    m = s.fov.model()
    rng = np.random.RandomState(5)
    n = 2_000_000
    x = (rng.rand(n).astype(np.float32) * 3 - 1) * np.float32(s.w)  # -w .. 2w
    y = (rng.rand(n).astype(np.float32) * 3 - 1) * np.float32(s.h)
    x[:1000] = np.float32(m.out_calib[2] * m.out_w) - np.float32(0.5) + (rng.rand(1000).astype(np.float32) - 0.5) * np.float32(1e-3)  # r ~ 0
    y[:1000] = np.float32(m.out_calib[3] * m.out_h) - np.float32(0.5) + (rng.rand(1000).astype(np.float32) - 0.5) * np.float32(1e-3)
    x[1000:2000] *= np.float32(1e4)  # huge radii: atanf's far branch
    x[2000:3000] *= np.float32(1e9)  # beyond 2^25, where glibc's atanf returns pi/2 outright (other fdlibm descendants: from 2^26 / 2^34)
    y[3000:3010] = np.float32(np.inf)
    x[3010:3020] = np.float32(np.nan)
    want_x, want_y = x.copy(), y.copy()
    oracle.distort(s.cam, np.array(list(m.out_calib), np.float32), want_x, want_y)

    def same(a, b):
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))

    # the drop-in class method: 2 * 10^6 points are a bulk call and run on the device (n >= 65536, fov_undistorter.cpp) ...
    cls_x, cls_y = x.copy(), y.copy()
    s.fov.distort_coordinates(cls_x, cls_y)
    assert same(cls_x, want_x) and same(cls_y, want_y)
    # ... the same points in pieces below the threshold stay on the host: both ways the reference's bits
    cls_x, cls_y = x.copy(), y.copy()
    for a in range(0, 200_000, 50_000):
        s.fov.distort_coordinates(cls_x[a:a + 50_000], cls_y[a:a + 50_000])
    assert same(cls_x[:200_000], want_x[:200_000]) and same(cls_y[:200_000], want_y[:200_000])
    got_x, got_y = x.copy(), y.copy()
    s.ctx.distort_points_host(m, got_x, got_y)
    bad = int((got_x.view(np.uint32) != want_x.view(np.uint32)).sum() + (got_y.view(np.uint32) != want_y.view(np.uint32)).sum())
    assert bad == 0, (name, bad)
    # device-pointer flavour
    torch = torch_cuda
    d_x, d_y = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    s.ctx.distort_points_device(m, d_x.data_ptr(), d_y.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_x.cpu().numpy().view(np.uint32), want_x.view(np.uint32))
    assert np.array_equal(d_y.cpu().numpy().view(np.uint32), want_y.view(np.uint32))


def test_pyramid(torch_cuda, oracle):
    """4-level box pyramid (own definition, parity unpinned by the reference)."""
    from mono_dataset_code_amd import capi

    torch = torch_cuda
    ctx = capi.Context(0)
    w, h, n = 96, 80, 3
    rng = np.random.RandomState(1)
    base = rng.rand(n, w * h).astype(np.float32) * 255
    base[0, 100] = np.nan
    d_base = torch.from_numpy(base).cuda()
    lv = [torch.empty(n * (w >> l) * (h >> l), dtype=torch.float32, device="cuda") for l in (1, 2, 3)]
    ctx.pyramid_batch(d_base.data_ptr(), w, h, 4, [t.data_ptr() for t in lv], n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for f in range(n):
        src, cw, ch = base[f], w, h
        for l in range(3):
            want = oracle.pyramid_level(src, cw, ch)
            got = lv[l].view(n, -1)[f].cpu().numpy()
            assert bits_equal(got, want), (f, l)
            src, cw, ch = want, cw // 2, ch // 2


@pytest.mark.parametrize("name", ["pyr_whole_black", "upsample", "small_explicit"])
def test_process_pyramid_fused(name, setups, oracle, torch_cuda):
    """getImage + pyramid in one call (config 5): levels 1..3 out of the remap kernel where the output is
    made of whole tiles, one pass per level otherwise; both must equal the oracle's level chain bit for bit."""
    from mono_dataset_code_amd import capi

    torch = torch_cuda
    s = setups(name)
    frames = np.stack(make_frames(s.W, s.H, n_noise=4))  # incl. saturated blobs -> NaN regions
    n = len(frames)
    flags = capi.RECTIFY | capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED
    base_want = [s.want(oracle, f, 1, 1, 1, 1) for f in frames]
    d_in = torch.from_numpy(frames).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for rows, levels, fpb, il, cols in ((32, 4, 0, 0, 64), (16, 4, 3, 1, 64), (64, 3, 0, 0, 64), (60, 4, 0, 0, 64), (32, 5, 2, 1, 64), (32, 1, 0, 0, 64),
                                        (64, 4, 2, 1, 64), (32, 4, 0, 0, 128), (16, 4, 2, 1, 128), (32, 3, 3, 0, 128)):
        s.ctx.set_option(capi.OPT_TILE_COLS, cols)
        s.ctx.set_option(capi.OPT_TILE_ROWS, rows)
        s.ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, fpb)
        s.ctx.set_option(capi.OPT_FRAME_INTERLEAVE, il)
        d_base = torch.full((n, s.w * s.h), -7.0, dtype=torch.float32, device="cuda")
        lv = [torch.full((n * (s.w >> l) * (s.h >> l),), -7.0, dtype=torch.float32, device="cuda") for l in range(1, levels)]
        s.ctx.process_pyramid_batch(d_in.data_ptr(), d_base.data_ptr(), levels, [t.data_ptr() for t in lv], n, flags, st)
        torch.cuda.synchronize()
        for f in range(n):
            assert bits_equal(d_base[f].cpu().numpy(), base_want[f]), (name, rows, levels, f)
            src, cw, ch = base_want[f], s.w, s.h
            for l in range(levels - 1):
                want = oracle.pyramid_level(src, cw, ch)
                assert bits_equal(lv[l].view(n, -1)[f].cpu().numpy(), want), (name, rows, cols, levels, f, l + 1)
                src, cw, ch = want, cw // 2, ch // 2
    s.ctx.set_option(capi.OPT_TILE_COLS, 0)
    s.ctx.set_option(capi.OPT_TILE_ROWS, 0)
    s.ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, 0)


def _two_stage_sweep(s, name, oracle, torch, selector, configs, kernel_name):
    """Shared body of the two-stage tests: every photometric flag combination, ragged batches, frames-per-workgroup /
    interleave variants, with and without the fused pyramid -- all bit-equal to the oracle, and the kernel that ran is
    the one asked for.  configs: (tile cols, tile rows, window buffers) triples; returns how many were plannable."""
    from mono_dataset_code_amd import capi

    big = name.startswith("full_")
    frames = np.stack(make_frames(s.W, s.H, n_noise=1 if big else 6))
    n = len(frames)
    d_in = torch.from_numpy(frames).cuda()
    st = torch.cuda.current_stream().cuda_stream
    s.ctx.set_option(capi.OPT_TWO_STAGE, selector)
    ran = 0
    try:
        for cols, rows, nbuf in configs:
            s.ctx.set_option(capi.OPT_TILE_COLS, cols)
            s.ctx.set_option(capi.OPT_TILE_ROWS, rows)
            s.ctx.set_option(capi.OPT_WINDOW_BUFFERS, nbuf)
            info = s.ctx.info()
            if not (info.tiled and info.two_stage == 1):
                continue  # windows too large for this kernel: another one keeps the job
            assert kernel_name in s.ctx.describe_launch(15, 0)
            ran += 1
            combos = [(1, 1, 1), (0, 0, 1)] if big else list(itertools.product((0, 1), repeat=3))
            for g, v, o in combos:
                flags = capi.RECTIFY | (capi.GAMMA * g) | (capi.VIGNETTE * v) | (capi.KILL_OVEREXPOSED * o)
                want = [s.want(oracle, f, 1, g, v, o) for f in frames]
                for m, fpb, il in ((n, 0, 0), (1, 0, 0), (n, 3, 1), (n, 2, 0)):
                    s.ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, fpb)
                    s.ctx.set_option(capi.OPT_FRAME_INTERLEAVE, il)
                    d_out = torch.full((m, s.w * s.h), -7.0, dtype=torch.float32, device="cuda")
                    s.ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), m, flags, st)
                    torch.cuda.synchronize()
                    got = d_out.cpu().numpy()
                    for f in range(m):
                        assert bits_equal(got[f], want[f]), (name, selector, cols, rows, nbuf, g, v, o, m, fpb, il, f)
            # fused pyramid (whole tiles only; otherwise the per-level passes run and must agree too), levels 4, 3, 2
            for levels, fpb in ((4, 2), (3, 0), (2, 3)):
                s.ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, fpb)
                s.ctx.set_option(capi.OPT_FRAME_INTERLEAVE, 0)
                d_base = torch.full((n, s.w * s.h), -7.0, dtype=torch.float32, device="cuda")
                lv = [torch.full((n * (s.w >> l) * (s.h >> l),), -7.0, dtype=torch.float32, device="cuda") for l in range(1, levels)]
                s.ctx.process_pyramid_batch(d_in.data_ptr(), d_base.data_ptr(), levels, [t.data_ptr() for t in lv], n, 15, st)
                torch.cuda.synchronize()
                for f in range(n):
                    src, cw, ch = s.want(oracle, frames[f], 1, 1, 1, 1), s.w, s.h
                    assert bits_equal(d_base[f].cpu().numpy(), src), (name, selector, cols, rows, nbuf, levels, f)
                    for l in range(levels - 1):
                        src = oracle.pyramid_level(src, cw, ch)
                        cw, ch = cw // 2, ch // 2
                        assert bits_equal(lv[l].view(n, -1)[f].cpu().numpy(), src), (name, selector, cols, rows, nbuf, levels, f, l + 1)
    finally:
        for o in (capi.OPT_TWO_STAGE, capi.OPT_TILE_COLS, capi.OPT_TILE_ROWS, capi.OPT_WINDOW_BUFFERS, capi.OPT_FRAMES_PER_BLOCK,
                  capi.OPT_FRAME_INTERLEAVE):
            s.ctx.set_option(o, 0)
    return ran


@pytest.mark.parametrize("name", ["mag4_full_black", "mag4_ragged", "upsample", "full_1280_to_1280"])
def test_strip_kernel(name, setups, oracle, torch_cuda):
    """remap_strip_kernel (wave-private 128 x 8 strips: own window, converted once per source pixel, 16 outputs per lane,
    pyramid levels out of registers), forced on with one to four u8 windows per wave (0 to 3 frames staged ahead)."""
    ran = _two_stage_sweep(setups(name), name, oracle, torch_cuda, 1, ((0, 0, 2), (0, 0, 1), (0, 0, 3), (0, 0, 4)), "remap_strip_kernel")
    assert ran == 4, "the strip kernel should be plannable for %s" % name


def test_strip_kernel_in_prefetched_chunks(setups, oracle, torch_cuda):
    """Large batches on the strip path go in chunks with a linear prefetch of the next chunk's source rows
    (MDC_OPT_PREFETCH_CHUNK): forced to 2 / 3 / 4 frames per chunk on a 9-frame batch (chunk sizes that do and do not
    divide it), the chunks on one stream and alternating over two (MDC_OPT_PREFETCH_STREAMS: the second one is internal
    and joined back -- the caller's stream order must hold: the results are read right after on the caller's stream),
    with and without the fused pyramid -- the same bits as one launch."""
    from mono_dataset_code_amd import capi

    torch = torch_cuda
    for name in ("mag4_full_black", "upsample"):
        s = setups(name)
        frames = np.stack(make_frames(s.W, s.H, n_noise=6))
        n = len(frames)
        d_in = torch.from_numpy(frames).cuda()
        st = torch.cuda.current_stream().cuda_stream
        want = [s.want(oracle, f, 1, 1, 1, 1) for f in frames]
        try:
            for chunk, streams in ((2, 2), (3, 2), (4, 0), (2, 1), (3, 1), (-1, 0)):
                s.ctx.set_option(capi.OPT_PREFETCH_CHUNK, chunk)
                s.ctx.set_option(capi.OPT_PREFETCH_STREAMS, streams)
                i = s.ctx.info()
                assert i.prefetch_streams == (0 if i.prefetch_chunk == 0 else 1 if streams == 1 else 2)
                assert (i.prefetch_chunk == 0) == (chunk < 0), 'the cameras of this test have a source box worth prefetching'
                d_out = torch.full((n, s.w * s.h), -7.0, dtype=torch.float32, device="cuda")
                s.ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), n, 15, st)
                levels = 4
                d_base = torch.full((n, s.w * s.h), -7.0, dtype=torch.float32, device="cuda")
                lv = [torch.full((n * (s.w >> l) * (s.h >> l),), -7.0, dtype=torch.float32, device="cuda") for l in range(1, levels)]
                s.ctx.process_pyramid_batch(d_in.data_ptr(), d_base.data_ptr(), levels, [t.data_ptr() for t in lv], n, 15, st)
                torch.cuda.synchronize()
                for f in range(n):
                    assert bits_equal(d_out[f].cpu().numpy(), want[f]), (name, chunk, f)
                    src, cw, ch = want[f], s.w, s.h
                    assert bits_equal(d_base[f].cpu().numpy(), src), (name, chunk, f)
                    for l in range(levels - 1):
                        src = oracle.pyramid_level(src, cw, ch)
                        cw, ch = cw // 2, ch // 2
                        assert bits_equal(lv[l].view(n, -1)[f].cpu().numpy(), src), (name, chunk, f, l + 1)
        finally:
            s.ctx.set_option(capi.OPT_PREFETCH_CHUNK, 0)
            s.ctx.set_option(capi.OPT_PREFETCH_STREAMS, 0)


def test_two_stage_is_chosen_by_source_pixels_per_output(setups):
    """Automatic choice: the scale-1 rectification of config 5 and magnifying remaps run on wave-private strips (fewer
    staged source pixels than outputs), the 1.5x downscale of the headline camera stays on the direct kernel."""
    assert setups("full_1280_to_1280").ctx.info().two_stage == 1
    assert "remap_strip_kernel<true, true" in setups("full_1280_to_1280").ctx.describe_launch(15, 4)
    assert setups("upsample").ctx.info().two_stage == 1
    assert setups("mag4_full_black").ctx.info().two_stage == 1
    assert setups("mag_full_black").ctx.info().two_stage == 0  # windows of some 128 x 8 tiles exceed 128 chunks
    assert setups("small_pinhole").ctx.info().two_stage == 0  # windows too large
    assert setups("full_1280_to_640").ctx.info().two_stage == 0
    assert "remap_tiled_kernel" in setups("full_1280_to_640").ctx.describe_launch(15, 0)


def test_pyramid_config5_full_size(setups, oracle, torch_cuda):
    """BASELINE.json configs[4] at its real geometry: 1280x1024 raw -> 1280x1024 rectified base + levels 1..3
    (640x512, 320x256, 160x128) out of ONE launch, every level of every checked frame bit-equal to the oracle's
    chain; then the stand-alone level passes on the same base must reproduce the fused levels on the whole batch."""
    from mono_dataset_code_amd import capi, synth

    torch = torch_cuda
    s = setups("full_1280_to_1280")
    info = s.ctx.info()
    assert info.tiled and (info.out_w, info.out_h) == (1280, 1024)
    n, levels = 12, 4
    npix, nout = s.W * s.H, s.w * s.h
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.empty(n * npix, dtype=torch.uint8, device="cuda")
    s.ctx.synth_frames(d_in.data_ptr(), 7, n, npix, synth.SEED, st)
    torch.cuda.synchronize()
    frames = d_in.view(n, npix).cpu().numpy().copy()
    frames[3] = synth.smooth_frame(s.W, s.H, 0.4)  # saturated blobs: contiguous NaN regions through every level
    d_in.copy_(torch.from_numpy(frames).reshape(-1))
    flags = capi.RECTIFY | capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED
    d_base = torch.full((n, nout), -7.0, dtype=torch.float32, device="cuda")
    lv = [torch.full((n, (s.w >> l) * (s.h >> l)), -7.0, dtype=torch.float32, device="cuda") for l in range(1, levels)]
    s.ctx.process_pyramid_batch(d_in.data_ptr(), d_base.data_ptr(), levels, [t.data_ptr() for t in lv], n, flags, st)
    torch.cuda.synchronize()
    for f in (0, 3, n - 1):
        src, cw, ch = s.want(oracle, frames[f], 1, 1, 1, 1), s.w, s.h
        assert bits_equal(d_base[f].cpu().numpy(), src), f
        for l in range(levels - 1):
            src = oracle.pyramid_level(src, cw, ch)
            cw, ch = cw // 2, ch // 2
            assert src.size == cw * ch == lv[l].shape[1]
            assert bits_equal(lv[l][f].cpu().numpy(), src), (f, l + 1)
    # whole batch: per-level passes over the fused base == the fused levels
    lv2 = [torch.full_like(t, -9.0) for t in lv]
    s.ctx.pyramid_batch(d_base.data_ptr(), s.w, s.h, levels, [t.data_ptr() for t in lv2], n, st)
    torch.cuda.synchronize()
    for a, b in zip(lv, lv2):
        assert bits_equal(a.cpu().numpy(), b.cpu().numpy())


def test_dso_gradients_of_pyramid_levels(setups, oracle, torch_cuda):
    """DSO hand-off (row f4; not in the reference, own definition, parity unpinned): (I, dx, dy) + absSquaredGrad of
    every level of the fused pyramid == the oracle's restatement, bit for bit, incl. NaN regions (non-finite
    differences -> 0) and odd sizes."""
    from mono_dataset_code_amd import capi

    torch = torch_cuda
    for name in ("pyr_whole_black", "ragged"):
        s = setups(name)
        frames = np.stack(make_frames(s.W, s.H, n_noise=2))
        n, levels = len(frames), 4
        d_in = torch.from_numpy(frames).cuda()
        st = torch.cuda.current_stream().cuda_stream
        d_base = torch.empty((n, s.w * s.h), dtype=torch.float32, device="cuda")
        lv = [torch.empty(n * (s.w >> l) * (s.h >> l), dtype=torch.float32, device="cuda") for l in range(1, levels)]
        s.ctx.process_pyramid_batch(d_in.data_ptr(), d_base.data_ptr(), levels, [t.data_ptr() for t in lv], n, 15, st)
        for l, t in enumerate([d_base.view(-1)] + lv):
            w, h = s.w >> l, s.h >> l
            d_dI = torch.full((n * w * h * 3,), -7.0, dtype=torch.float32, device="cuda")
            d_abs = torch.full((n * w * h,), -7.0, dtype=torch.float32, device="cuda")
            s.ctx.gradients_batch(t.data_ptr(), w, h, d_dI.data_ptr(), d_abs.data_ptr(), n, st)
            torch.cuda.synchronize()
            src = t.cpu().numpy().reshape(n, w * h)
            for f in range(n):
                dI, a = oracle.gradients(src[f], w, h)
                assert bits_equal(d_dI.view(n, -1)[f].cpu().numpy(), dI), (name, l, f)
                assert bits_equal(d_abs.view(n, -1)[f].cpu().numpy(), a), (name, l, f)


@pytest.mark.parametrize("name", ["pyr_whole_black", "ragged", "upsample", "mag4_full_black"])
def test_dso_preprocessing_in_one_call(name, setups, oracle, torch_cuda):
    """mdc_process_pyramid_gradients_batch_device (row f4; parity unpinned: DSO is not in the reference, own definition and oracle):
    base, levels and the gradient images of every level from one call, walked in chunks (also a chunk size that does not divide
    the batch) -- every array bit-equal to the oracle's chain remap -> box levels -> orc_gradients."""
    from mono_dataset_code_amd import capi

    torch = torch_cuda
    s = setups(name)
    frames = np.stack(make_frames(s.W, s.H, n_noise=3))
    n = len(frames)
    d_in = torch.from_numpy(frames).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for levels, chunk in ((4, 0), (4, 2), (3, 1), (1, 4), (5, 3)):
        if (min(s.w, s.h) >> (levels - 1)) < 1:
            continue
        dims = [(s.w >> l, s.h >> l) for l in range(levels)]
        d_base = torch.full((n, s.w * s.h), -7.0, dtype=torch.float32, device="cuda")
        lv = [torch.full((n, w * h), -7.0, dtype=torch.float32, device="cuda") for w, h in dims[1:]]
        dI = [torch.full((n, w * h * 3), -7.0, dtype=torch.float32, device="cuda") for w, h in dims]
        ab = [torch.full((n, w * h), -7.0, dtype=torch.float32, device="cuda") for w, h in dims]
        s.ctx.process_pyramid_gradients_batch(d_in.data_ptr(), d_base.data_ptr(), levels, [t.data_ptr() for t in lv], [t.data_ptr() for t in dI],
                                              [t.data_ptr() for t in ab], n, 15, chunk, st)
        torch.cuda.synchronize()
        for f in range(n):
            src = s.want(oracle, frames[f], 1, 1, 1, 1)
            for l, (w, h) in enumerate(dims):
                if l:
                    src = oracle.pyramid_level(src, dims[l - 1][0], dims[l - 1][1])
                got = (d_base if l == 0 else lv[l - 1])[f].cpu().numpy()
                assert bits_equal(got, src), (name, levels, chunk, f, l)
                want_dI, want_abs = oracle.gradients(src, w, h)
                assert bits_equal(dI[l][f].cpu().numpy(), want_dI), (name, levels, chunk, f, l, "dI")
                assert bits_equal(ab[l][f].cpu().numpy(), want_abs), (name, levels, chunk, f, l, "abs2")


@pytest.mark.parametrize("name", ["small_crop", "ragged"])
def test_process_frames_host_pipeline(name, setups, oracle):
    """Many host frames in one call (chunks on two streams): equal to the oracle frame by frame, with
    pageable and with page-locked buffers, for counts around the chunk size and all flag modes that matter."""
    from mono_dataset_code_amd import capi

    s = setups(name)
    base = make_frames(s.W, s.H, n_noise=5)
    frames = [base[i % len(base)] for i in range(37)]
    for rect, g, v, o in ((1, 1, 1, 1), (0, 1, 1, 0), (1, 0, 0, 0)):
        flags = (capi.RECTIFY * rect) | (capi.GAMMA * g) | (capi.VIGNETTE * v) | (capi.KILL_OVEREXPOSED * o)
        n_out = s.w * s.h if rect else s.W * s.H
        want = [s.want(oracle, f, rect, g, v, o) for f in base]
        for n in (0, 1, 16, 17, 37):
            outs = [np.full(n_out, -7.0, np.float32) for _ in range(n)]
            s.ctx.process_frames_host(frames[:n], outs, flags)
            for i in range(n):
                assert bits_equal(outs[i], want[i % len(base)]), (name, rect, g, v, o, n, i)
    # page-locked buffers from the library's allocator
    n = 20
    pin_in = capi.PinnedArray((n, s.W * s.H), np.uint8)
    pin_out = capi.PinnedArray((n, s.w * s.h), np.float32)
    for i in range(n):
        pin_in.array[i] = frames[i]
    pin_out.array[:] = -7.0
    s.ctx.process_frames_host([pin_in.array[i] for i in range(n)], [pin_out.array[i] for i in range(n)], 15)
    want = [s.want(oracle, f, 1, 1, 1, 1) for f in base]
    for i in range(n):
        assert bits_equal(pin_out.array[i], want[i % len(base)]), (name, "pinned", i)


@pytest.mark.parametrize("name", ["small_crop", "ragged", "mag4_full_black"])
def test_zero_copy_host_calls(name, setups, oracle):
    """MDC_OPT_ZERO_COPY: buffers in mapped page-locked memory are read / written by the kernels directly -- every host entry
    point, every mix of page-locked and pageable sides, buffers that start inside an allocation, frames scattered over the
    block (runs of 1, 2, 3 back-to-back frames, a step backwards), against the oracle, with zero copy on and switched off."""
    from mono_dataset_code_amd import capi

    s = setups(name)
    base = make_frames(s.W, s.H, n_noise=5)
    npi, npo = s.W * s.H, s.w * s.h
    n = 9
    pin_in = capi.PinnedArray((n + 1, npi), np.uint8)
    pin_out = capi.PinnedArray((n + 1, npo), np.float32)
    pin_unm = capi.PinnedArray((4, npi), np.float32)
    for i in range(n + 1):
        pin_in.array[i] = base[i % len(base)]
    want = [s.want(oracle, f, 1, 1, 1, 1) for f in base]
    want_unmap = [oracle.unmap(f, s.ginv, s.vinv, True, True, 1, 1, 1) for f in base]
    want_und8 = [oracle.undistort(f, s.rx, s.ry, s.W) for f in base]
    try:
        for zc in (0, 2, 1):
            s.ctx.set_option(capi.OPT_ZERO_COPY, zc)
            # single-frame calls: pinned -> pinned, pageable -> pinned, pinned -> pageable
            for src_pinned, dst_pinned in ((1, 1), (0, 1), (1, 0)):
                for i in (0, 3):
                    raw = pin_in.array[i] if src_pinned else base[i % len(base)].copy()
                    out = pin_out.array[i] if dst_pinned else np.empty(npo, np.float32)
                    out[:] = -7.0
                    s.ctx.process_host(raw, out, 15)
                    assert bits_equal(out, want[i % len(base)]), (name, zc, "process_host", src_pinned, dst_pinned, i)
                    out_u = pin_unm.array[i] if dst_pinned else np.empty(npi, np.float32)
                    out_u[:] = -7.0
                    s.ctx.unmap_host(raw, out_u, 7)
                    assert bits_equal(out_u, want_unmap[i % len(base)]), (name, zc, "unmap_host", src_pinned, dst_pinned, i)
                    out[:] = -7.0
                    s.ctx.undistort_host(raw, out)
                    assert bits_equal(out, want_und8[i % len(base)]), (name, zc, "undistort_u8", src_pinned, dst_pinned, i)
                    if src_pinned and dst_pinned:  # undistort<float> from a page-locked float frame into a pageable result
                        res = np.full(npo, -7.0, np.float32)
                        s.ctx.undistort_host(out_u, res)
                        assert bits_equal(res, oracle.undistort(want_unmap[i % len(base)], s.rx, s.ry, s.W)), (name, zc, "undistort_f32", i)
            # many frames: rows of the page-locked blocks in an order with runs of 1, 2 and 3 neighbours and a step backwards
            order = [0, 1, 2, 4, 6, 7, 5, 3, 8]
            ins = [pin_in.array[i] for i in order]
            outs = [pin_out.array[i] for i in order]
            pin_out.array[:] = -7.0
            s.ctx.process_frames_host(ins, outs, 15)
            for k, i in enumerate(order):
                assert bits_equal(outs[k], want[i % len(base)]), (name, zc, "frames_host", k, i)
            # results into page-locked rows, frames from pageable memory; and the other way round
            pin_out.array[:] = -7.0
            s.ctx.process_frames_host([base[i % len(base)].copy() for i in order], outs, 15)
            for k, i in enumerate(order):
                assert bits_equal(outs[k], want[i % len(base)]), (name, zc, "frames_host pageable in", k, i)
            loose = [np.full(npo, -7.0, np.float32) for _ in order]
            s.ctx.process_frames_host(ins, loose, 15)
            for k, i in enumerate(order):
                assert bits_equal(loose[k], want[i % len(base)]), (name, zc, "frames_host pageable out", k, i)
    finally:
        s.ctx.set_option(capi.OPT_ZERO_COPY, 0)


@pytest.mark.parametrize("name", ["small_crop", "upsample"])
def test_device_batch_calls_inside_a_hip_graph(name, setups, oracle, torch_cuda):
    """The device-batch entry points only enqueue on the caller's stream: captured into a HIP graph (torch.cuda.CUDAGraph) and
    replayed on fresh input they give the oracle's results -- the fused pass, the pyramid (on the strip path the chunks stay on
    the capturing stream: a shared second stream must not be drawn into a capture) and the gradients."""
    from mono_dataset_code_amd import capi

    torch = torch_cuda
    s = setups(name)
    frames = np.stack(make_frames(s.W, s.H, n_noise=7))
    n = len(frames)
    levels = 3
    d_in = torch.zeros((n, s.W * s.H), dtype=torch.uint8, device="cuda")
    d_out = torch.full((n, s.w * s.h), -7.0, dtype=torch.float32, device="cuda")
    d_base = torch.full((n, s.w * s.h), -7.0, dtype=torch.float32, device="cuda")
    lv = [torch.full((n * (s.w >> l) * (s.h >> l),), -7.0, dtype=torch.float32, device="cuda") for l in range(1, levels)]
    try:
        s.ctx.set_option(capi.OPT_PREFETCH_CHUNK, 2)  # (strip cameras: chunks + prefetch launches inside the capture too)
        # first call outside any capture: lazy attribute settings and plan uploads are not stream operations
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            s.ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), n, 15, st.cuda_stream)
            s.ctx.process_pyramid_batch(d_in.data_ptr(), d_base.data_ptr(), levels, [t.data_ptr() for t in lv], n, 15, st.cuda_stream)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            s.ctx.process_batch(d_in.data_ptr(), d_out.data_ptr(), n, 15, st.cuda_stream)
            s.ctx.process_pyramid_batch(d_in.data_ptr(), d_base.data_ptr(), levels, [t.data_ptr() for t in lv], n, 15, st.cuda_stream)
        for rep in range(2):
            batch = np.roll(frames, rep + 1, axis=0)
            d_in.copy_(torch.from_numpy(batch).cuda())
            d_out.fill_(-7.0)
            d_base.fill_(-7.0)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            for f in range(n):
                want = s.want(oracle, batch[f], 1, 1, 1, 1)
                assert bits_equal(d_out[f].cpu().numpy(), want), (name, rep, f)
                assert bits_equal(d_base[f].cpu().numpy(), want), (name, rep, f, "base")
                src, cw, ch = want, s.w, s.h
                for l in range(levels - 1):
                    src = oracle.pyramid_level(src, cw, ch)
                    cw, ch = cw // 2, ch // 2
                    assert bits_equal(lv[l].view(n, -1)[f].cpu().numpy(), src), (name, rep, f, l + 1)
    finally:
        s.ctx.set_option(capi.OPT_PREFETCH_CHUNK, 0)


def test_table_blob_roundtrip(setups, oracle, torch_cuda):
    """export -> import into a second context (what the RCCL broadcast carries)."""
    from mono_dataset_code_amd import capi

    s = setups("small_crop")
    blob = s.ctx.export_tables()
    c2 = capi.Context(0)
    c2.import_tables(blob)
    assert np.array_equal(c2.export_tables(), blob)
    raw = make_frames(s.W, s.H)[0]
    a, b = np.zeros(s.w * s.h, np.float32), np.zeros(s.w * s.h, np.float32)
    s.ctx.process_host(raw, a, 15)
    c2.process_host(raw, b, 15)
    assert bits_equal(a, b)


def test_error_codes(setups):
    from mono_dataset_code_amd import capi

    c = capi.Context(0)
    out = np.zeros(10, np.float32)
    assert c.undistort_host(np.zeros(10, np.uint8), out, check=False) == capi.ERR_STATE
    assert c.process_host(np.zeros(10, np.uint8), out, capi.RECTIFY, check=False) == capi.ERR_STATE
    s = setups("small_explicit")
    assert s.ctx.undistort_host(np.zeros(10, np.uint8), out, check=False) == capi.ERR_SIZE
    assert "wrong input image" in s.ctx.last_error()
    with pytest.raises(capi.MdcError):
        c.set_remap(np.array([5.0], np.float32), np.array([500.0], np.float32), 16, 16, 1, 1)


def test_tune_placement_picks_among_candidate_buffers(setups, oracle, torch_cuda):
    """mdc_tune_placement_device (include/mdc_hip.h): the fused pass timed on candidate frame / result buffers; the indices returned are
    valid, every candidate has a time, every result candidate holds the pass's results afterwards (bit-exact), bad arguments are refused."""
    import ctypes

    from mono_dataset_code_amd import capi, synth

    torch = torch_cuda
    s = setups("full_1280_to_640")
    n, npix, nout = 48, s.W * s.H, s.w * s.h
    st = torch.cuda.current_stream().cuda_stream
    flags = capi.RECTIFY | capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED
    ins = [torch.empty(n * npix, dtype=torch.uint8, device="cuda") for _ in range(3)]
    outs = [torch.full((n * nout,), -5.0, dtype=torch.float32, device="cuda") for _ in range(4)]
    for t in ins:
        s.ctx.synth_frames(t.data_ptr(), 11, n, npix, synth.SEED, st)
    bi, bo, ms = s.ctx.tune_placement([t.data_ptr() for t in ins], [t.data_ptr() for t in outs], n, flags, st)
    assert 0 <= bi < 3 and 0 <= bo < 4 and len(ms) == 3 and all(len(r) == 4 for r in ms)
    flat = [x for r in ms for x in r]
    assert all(0 < x < 50 for x in flat) and ms[bi][bo] == min(flat)
    frames = ins[0].view(n, npix)[:2].cpu().numpy()
    for t in outs:  # every result candidate was written by the pass
        got = t.view(n, nout)[:2].cpu().numpy()
        for f in range(2):
            assert bits_equal(got[f], s.want(oracle, frames[f], 1, 1, 1, 1))
    with pytest.raises(capi.MdcError):
        s.ctx.tune_placement([], [outs[0].data_ptr()], n, flags, st)
    with pytest.raises(capi.MdcError):
        s.ctx.tune_placement([ins[0].data_ptr()], [0], n, flags, st)


@pytest.mark.parametrize("strategy", ["first", "malloc", "vmm"])
def test_alloc_placed_hands_out_working_buffers(setups, oracle, torch_cuda, strategy, monkeypatch):
    """mdc_alloc_placed_device (include/mdc_hip.h): a frame buffer and a result buffer made by the product's allocator -- plain, chosen among timed
    hipMalloc candidates, assembled from classified physical pieces -- hold the pass's bit-exact results like any other device memory, report what
    was done, and can be given back and made again (a few rounds: the assembled ranges map and unmap physical pieces)."""
    from mono_dataset_code_amd import capi, synth

    torch = torch_cuda
    s = setups("full_1280_to_640")
    monkeypatch.setenv("MDC_PLACE_CANDIDATES", "2")
    monkeypatch.setenv("MDC_PLACE_SPREAD_MB", "512")
    monkeypatch.setenv("MDC_PLACE_PIECE_MIB", "64")
    n, npix, nout = 600, s.W * s.H, s.w * s.h  # 0.79 GB of frames + 0.74 GB of results
    st = torch.cuda.current_stream().cuda_stream
    flags = capi.RECTIFY | capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED
    which = {"first": capi.PLACE_FIRST, "malloc": capi.PLACE_MALLOC, "vmm": capi.PLACE_VMM}[strategy]
    for rnd in range(3):
        b = s.ctx.alloc_placed(n, flags, which, st)
        d = b.describe()
        assert b.strategy == which and b.d_in and b.d_out and b.in_bytes >= n * npix and b.out_bytes >= n * nout * 4 and d["how"]
        if strategy == "malloc":
            m = d["ms_frames_i_results_j"]
            assert len(m) == 2 and all(0 < x < 50 for r in m for x in r) and m[d["picked_frames"]][d["picked_results"]] == min(x for r in m for x in r)
            assert d["ms_on_first_allocations"] == m[0][0]
        if strategy == "vmm":
            assert d["pieces"] >= 12 + 12 and d["piece_mib"] == 64 and sum(d["class_count"]) == d["pieces"] and d["ms_on_chosen_pair"] > 0
        s.ctx.synth_frames(b.d_in, 5 + rnd, n, npix, synth.SEED, st)
        s.ctx.process_batch(b.d_in, b.d_out, n, flags, st)
        torch.cuda.synchronize()
        for f in (0, n // 2 + 1, n - 1):  # (frames that lie in different pieces of an assembled range)
            raw = s.ctx.copy_to_host(b.d_in + f * npix, npix, np.uint8)
            assert np.array_equal(raw, synth.noise_frames(5 + rnd + f, 1, npix)[0])
            assert bits_equal(s.ctx.copy_to_host(b.d_out + f * nout * 4, nout, np.float32), s.want(oracle, raw, 1, 1, 1, 1))
        s.ctx.free_placed(b)
        assert not b.handle and not b.d_in
        s.ctx.free_placed(b)  # giving back twice is harmless
    small = s.ctx.alloc_placed(8, flags, capi.PLACE_AUTO, st)  # a pair that lives in the Infinity Cache: no search
    assert small.strategy == capi.PLACE_FIRST
    s.ctx.free_placed(small)
    with pytest.raises(capi.MdcError):
        s.ctx.alloc_placed(0, flags, which, st)
    with pytest.raises(capi.MdcError):
        s.ctx.alloc_placed(4, flags, which, st, in_bytes=100)  # smaller than four frames
    with pytest.raises(capi.MdcError):
        capi.Context(0).alloc_placed(4, flags, which, st)  # no tables: the pass cannot be probed


def test_striped_set_holds_a_pyramid_step(setups, oracle, torch_cuda, monkeypatch):
    """mdc_alloc_striped_set_device: the further outputs of a step (config 5's levels) from the allocator -- large buffers as ranges of their own,
    small ones sharing a range -- next to an assembled pair; the step's results on them equal the oracle's chain bit for bit; sets can be
    made and given back repeatedly; bad arguments are refused."""
    from mono_dataset_code_amd import capi, synth

    torch = torch_cuda
    s = setups("full_1280_to_1280")
    monkeypatch.setenv("MDC_PLACE_PIECE_MIB", "64")
    n, npix, nout = 96, s.W * s.H, s.w * s.h
    st = torch.cuda.current_stream().cuda_stream
    flags = capi.RECTIFY | capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED
    for rnd in range(2):
        pair = s.ctx.alloc_placed(n, flags, capi.PLACE_VMM, st)
        sizes = [n * (s.w >> l) * (s.h >> l) * 4 for l in (1, 2, 3)]  # 126 MB (a range of its own), 31 MB and 8 MB (sharing one)
        lv = s.ctx.alloc_striped_set(sizes, st)
        assert lv.n == 3 and lv.strategy == capi.PLACE_VMM and all(lv.d_ptr[k] for k in range(3)) and b"striped" in lv.note
        assert lv.d_ptr[2] - lv.d_ptr[1] == (sizes[1] + (2 << 20) - 1) // (2 << 20) * (2 << 20)  # the two small ones back to back at a 2-MiB boundary
        s.ctx.synth_frames(pair.d_in, 40 + rnd, n, npix, synth.SEED, st)
        s.ctx.process_pyramid_batch(pair.d_in, pair.d_out, 4, [lv.d_ptr[k] for k in range(3)], n, flags, st)
        torch.cuda.synchronize()
        for f in (0, n - 1):
            raw = synth.noise_frames(40 + rnd + f, 1, npix)[0]
            src, cw, ch = s.want(oracle, raw, 1, 1, 1, 1), s.w, s.h
            assert bits_equal(s.ctx.copy_to_host(pair.d_out + f * nout * 4, nout, np.float32), src)
            for l in range(3):
                src = oracle.pyramid_level(src, cw, ch)
                cw, ch = cw // 2, ch // 2
                assert bits_equal(s.ctx.copy_to_host(lv.d_ptr[l] + f * cw * ch * 4, cw * ch, np.float32), src), (rnd, f, l + 1)
        s.ctx.free_striped_set(lv)
        s.ctx.free_placed(pair)
        assert not lv.handle
    with pytest.raises(capi.MdcError):
        s.ctx.alloc_striped_set([], st)
    with pytest.raises(capi.MdcError):
        s.ctx.alloc_striped_set([1 << 20, 0], st)
    monkeypatch.setenv("MDC_PLACEMENT", "first")  # switched off: plain allocations, said so
    plain = s.ctx.alloc_striped_set([1 << 20, 3 << 20], st)
    assert plain.strategy == capi.PLACE_FIRST and plain.d_ptr[0] and plain.d_ptr[1]
    s.ctx.free_striped_set(plain)


def test_device_alloc_stripes_large_buffers(setups, oracle, torch_cuda):
    """mdc_device_alloc: a GiB or more comes striped over the device's memory classes (the same machinery as mdc_alloc_striped_set_device),
    smaller buffers from hipMalloc; both hold results like any device memory and go back through mdc_device_free."""
    from mono_dataset_code_amd import capi, synth

    torch = torch_cuda
    s = setups("full_1280_to_640")
    n, npix, nout = 900, s.W * s.H, s.w * s.h  # 1.18 GB of frames, 1.11 GB of results
    st = torch.cuda.current_stream().cuda_stream
    flags = capi.RECTIFY | capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED
    for rnd in range(2):
        d_in, d_out, small = s.ctx.device_alloc(n * npix), s.ctx.device_alloc(n * nout * 4), s.ctx.device_alloc(4096)
        assert d_in and d_out and small and len({d_in, d_out, small}) == 3
        s.ctx.synth_frames(d_in, 70 + rnd, n, npix, synth.SEED, st)
        s.ctx.process_batch(d_in, d_out, n, flags, st)
        torch.cuda.synchronize()
        for f in (0, n // 3, n - 1):
            raw = s.ctx.copy_to_host(d_in + f * npix, npix, np.uint8)
            assert np.array_equal(raw, synth.noise_frames(70 + rnd + f, 1, npix)[0])
            assert bits_equal(s.ctx.copy_to_host(d_out + f * nout * 4, nout, np.float32), s.want(oracle, raw, 1, 1, 1, 1))
        for p in (d_in, d_out, small):
            s.ctx.device_free(p)
