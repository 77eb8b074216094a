"""vignetteCalib solver loops (reference src/main_vignetteCalib.cpp:395-527).

CPU : oracle/mdc_oracle.c's restatement == the reference's own loop text (cut out of its main() at build time,
      oracle/vcal_extract.py -> oracle/_ref/libvcal_ref.so), bit for bit, over several alternating iterations.
GPU : mdc_vcal_plane_step == oracle bit for bit (each plane point sums over the images in the reference's order);
      mdc_vcal_vignette_step within 1e-5 relative (a scatter-add: the summation ORDER of the reference's sequential
      loop cannot be kept by concurrent atomics; float sums of positive terms differ in the last bits only);
      mdc_vcal_vignette_step_indexed (the scatter inverted into an ordered gather over mdc_vcal_index) == oracle bit
      for bit, so the whole alternating iteration on the GPU equals the reference's."""
import numpy as np
import pytest

from conftest import bits_equal


def problem(seed=0, n=6, wI=48, hI=40, gw=36, gh=30):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:hI, 0:wI].astype(np.float32)
    r2 = ((x - wI / 2) ** 2 + (y - hI / 2) ** 2) / (wI * hI / 4.0)
    vig_true = (1.0 - 0.5 * r2).astype(np.float32)
    gy, gx = np.mgrid[0:gh, 0:gw].astype(np.float32)
    plane_true = (60 + 40 * np.sin(0.4 * gx) * np.cos(0.3 * gy)).astype(np.float32)
    images = np.zeros((n, hI, wI), np.float32)
    p2x = np.zeros((n, gw * gh), np.float32)
    p2y = np.zeros((n, gw * gh), np.float32)
    for i in range(n):
        # an affine view of the plane that stays inside [1, w-2] x [1, h-2], different per image
        sx, sy = rng.uniform(0.8, 1.15), rng.uniform(0.8, 1.15)
        ox, oy = rng.uniform(1.5, 4.0), rng.uniform(1.5, 4.0)
        px = (ox + sx * gx + 0.05 * gy).astype(np.float32)
        py = (oy + sy * gy - 0.04 * gx).astype(np.float32)
        bad = (px < 1) | (py < 1) | (px > wI - 2.5) | (py > hI - 2.5) | (rng.random((gh, gw)) < 0.03)
        px[bad] = np.nan
        py[bad] = np.nan
        p2x[i], p2y[i] = px.reshape(-1), py.reshape(-1)
        # the image: plane colour seen through the vignette (nearest plane point per pixel, good enough for a test)
        ix = np.clip(np.round((x - ox) / sx), 0, gw - 1).astype(int)
        iy = np.clip(np.round((y - oy) / sy), 0, gh - 1).astype(int)
        img = plane_true[iy, ix] * vig_true + rng.normal(0, 0.5, (hI, wI)).astype(np.float32)
        img[rng.random((hI, wI)) < 0.01] = np.nan  # masked gradients (:294-300)
        images[i] = img
    return images, p2x, p2y, gw, gh


def test_oracle_equals_reference_loops_bit_for_bit(oracle):
    from oracle import loader

    try:
        ref = loader.VcalRef()
    except OSError as e:
        pytest.skip(str(e))
    for seed in range(3):
        images, p2x, p2y, gw, gh = problem(seed)
        n, hI, wI = images.shape
        pc_o = np.zeros(gw * gh, np.float32)  # the reference's `new float[]` is uninitialised; any start value is legal
        pc_r = pc_o.copy()
        vf_o = np.ones(hI * wI, np.float32)
        vf_r = vf_o.copy()
        for it in range(6):
            oth2 = 10000 * 10000 if it < 3 else 15 * 15  # :397-398
            pc_o, ff_o, fc_o, e_o, r_o = oracle.vcal_plane_step(images, p2x, p2y, pc_o, vf_o, oth2)
            pc_r, ff_r, fc_r, e_r, r_r = ref.plane_step(images, p2x, p2y, gw, gh, pc_r, vf_r, oth2)
            assert bits_equal(ff_o, ff_r) and bits_equal(fc_o, fc_r) and bits_equal(pc_o, pc_r), (seed, it, "plane")
            assert e_o == e_r and r_o == r_r and r_o > 100
            vf_o, tt_o, ct_o, e_o, r_o = oracle.vcal_vignette_step(images, p2x, p2y, pc_o, vf_o, oth2)
            vf_r, tt_r, ct_r, e_r, r_r = ref.vignette_step(images, p2x, p2y, gw, gh, pc_r, vf_r, oth2)
            assert bits_equal(tt_o, tt_r) and bits_equal(ct_o, ct_r) and bits_equal(vf_o, vf_r), (seed, it, "vignette")
            assert e_o == e_r and r_o == r_r
        assert np.nanmax(vf_o) == 1.0 and np.isnan(pc_o).sum() < gw * gh // 2


@pytest.mark.gpu
def test_gpu_steps_against_oracle(oracle):
    import torch

    from mono_dataset_code_amd import capi

    ctx = capi.Context(0)
    st = torch.cuda.current_stream().cuda_stream
    for seed, shape in ((0, {}), (1, dict(n=9, wI=160, hI=120, gw=140, gh=110))):
        images, p2x, p2y, gw, gh = problem(seed, **shape)
        n, hI, wI = images.shape
        d_img, d_x, d_y = (torch.from_numpy(a).cuda() for a in (images, p2x, p2y))
        pc = np.zeros(gw * gh, np.float32)
        vf = np.ones(hI * wI, np.float32)
        d_pc, d_vf = torch.from_numpy(pc).cuda(), torch.from_numpy(vf).cuda()
        for it in range(6):
            oth2 = 10000 * 10000 if it < 3 else 15 * 15
            pc_o, ff_o, fc_o, e_o, r_o = oracle.vcal_plane_step(images, p2x, p2y, pc, vf, oth2)
            ff, fc, e, r = ctx.vcal_plane_step(d_img, d_x, d_y, d_pc, d_vf, oth2, st)
            torch.cuda.synchronize()
            assert bits_equal(ff.cpu().numpy(), ff_o) and bits_equal(fc.cpu().numpy(), fc_o), (seed, it)
            assert bits_equal(d_pc.cpu().numpy(), pc_o) and r == r_o and abs(e - e_o) <= 1e-9 * abs(e_o)
            pc = pc_o
            vf_o, tt_o, ct_o, e_o, r_o = oracle.vcal_vignette_step(images, p2x, p2y, pc, vf, oth2)
            tt, ct, e, r = ctx.vcal_vignette_step(d_img, d_x, d_y, d_pc, d_vf, oth2, st)
            torch.cuda.synchronize()
            for got, want in ((tt.cpu().numpy(), tt_o), (ct.cpu().numpy(), ct_o), (d_vf.cpu().numpy(), vf_o)):
                assert np.array_equal(np.isnan(got), np.isnan(want))
                m = ~np.isnan(want)
                assert np.allclose(got[m], want[m], rtol=1e-5, atol=1e-6), (seed, it, np.abs(got[m] - want[m]).max())
            assert r == r_o and abs(e - e_o) <= 1e-6 * abs(e_o) + 1e-6
            # the next iteration continues from the GPU's own (tolerance-equal) factors on both sides
            vf = d_vf.cpu().numpy().copy()


@pytest.mark.gpu
def test_gpu_indexed_vignette_step_is_bit_identical(oracle):
    """The vignette half-iteration as an ordered gather over mdc_vcal_index: the whole alternating iteration run on
    the GPU (plane step, indexed vignette step) stays bit-identical to the oracle (== the reference's loops, test
    above) over 6 iterations WITHOUT ever re-synchronising the two sides; the atomic variant needs a tolerance."""
    import torch

    from mono_dataset_code_amd import capi

    ctx = capi.Context(0)
    st = torch.cuda.current_stream().cuda_stream
    for seed, shape in ((0, {}), (1, dict(n=9, wI=160, hI=120, gw=140, gh=110)), (2, dict(n=40, wI=97, hI=61, gw=150, gh=95))):
        images, p2x, p2y, gw, gh = problem(seed, **shape)
        if seed == 2:  # coordinates the reference's caller would have masked (:283-300): dropped, not written out of bounds
            p2x[3, 17], p2y[3, 17] = np.float32(images.shape[2] - 0.5), np.float32(5.0)
            p2x[4, 18], p2y[4, 18] = np.float32(5.0), np.float32(np.nan)
        n, hI, wI = images.shape
        d_img, d_x, d_y = (torch.from_numpy(a).cuda() for a in (images, p2x, p2y))
        index = ctx.vcal_index(d_img, d_x, d_y, st)
        valid = ~np.isnan(p2x) & ~np.isnan(p2y) & (p2x < wI - 1) & (p2y < hI - 1)
        assert 0 < index.entries <= 4 * int(valid.sum()) and index.bytes >= 16 * index.entries
        if seed == 2:
            p2x[3, 17] = p2x[4, 18] = np.nan  # what the oracle sees: the same samples masked by the caller
        pc = np.zeros(gw * gh, np.float32)
        vf = np.ones(hI * wI, np.float32)
        d_pc, d_vf = torch.from_numpy(pc).cuda(), torch.from_numpy(vf).cuda()
        for it in range(6):
            oth2 = 10000 * 10000 if it < 3 else 15 * 15
            pc, ff_o, fc_o, e_o, r_o = oracle.vcal_plane_step(images, p2x, p2y, pc, vf, oth2)
            if seed == 2:
                d_x2, d_y2 = torch.from_numpy(p2x).cuda(), torch.from_numpy(p2y).cuda()
                ctx.vcal_plane_step(d_img, d_x2, d_y2, d_pc, d_vf, oth2, st)
            else:
                ctx.vcal_plane_step(d_img, d_x, d_y, d_pc, d_vf, oth2, st)
            assert bits_equal(d_pc.cpu().numpy(), pc), (seed, it)
            vf, tt_o, ct_o, e_o, r_o = oracle.vcal_vignette_step(images, p2x, p2y, pc, vf, oth2)
            for rep in range(2):  # repeatable: no atomics on the data path
                d_vf_in = d_vf.clone()
                tt, ct, e, r = ctx.vcal_vignette_step_indexed(index, d_pc, d_vf_in, oth2, st)
                torch.cuda.synchronize()
                assert bits_equal(tt.cpu().numpy(), tt_o), (seed, it, rep)
                assert bits_equal(ct.cpu().numpy(), ct_o), (seed, it, rep)
                assert bits_equal(d_vf_in.cpu().numpy(), vf), (seed, it, rep)
                assert r == r_o and abs(e - e_o) <= 1e-9 * abs(e_o) + 1e-9
            d_vf = d_vf_in
        index.close()
    # no valid sample at all: empty lists, every factor NaN -- as the oracle says
    images, p2x, p2y, gw, gh = problem(5)
    p2x[:], p2y[:] = np.nan, np.nan
    n, hI, wI = images.shape
    d_img, d_x, d_y = (torch.from_numpy(a).cuda() for a in (images, p2x, p2y))
    index = ctx.vcal_index(d_img, d_x, d_y, st)
    assert index.entries == 0
    pc = np.full(gw * gh, 50.0, np.float32)
    vf_o, tt_o, ct_o, e_o, r_o = oracle.vcal_vignette_step(images, p2x, p2y, pc, np.ones(hI * wI, np.float32), 225)
    d_vf = torch.ones(hI * wI, dtype=torch.float32, device="cuda")
    tt, ct, e, r = ctx.vcal_vignette_step_indexed(index, torch.from_numpy(pc).cuda(), d_vf, 225, st)
    assert bits_equal(tt.cpu().numpy(), tt_o) and bits_equal(ct.cpu().numpy(), ct_o) and bits_equal(d_vf.cpu().numpy(), vf_o) and r == r_o == 0


@pytest.mark.gpu
def test_gpu_solver_loop_in_one_call(oracle):
    """mdc_vcal_solve_device = the reference's whole loop :395-527 (index, 2 x max_iterations half-iterations, outlier
    schedule) in one call: plane colours and vignette factors after 10 iterations are the oracle's, bit for bit; the
    printed statistics agree (R exactly, E to 1e-9)."""
    import torch

    from mono_dataset_code_amd import capi

    ctx = capi.Context(0)
    st = torch.cuda.current_stream().cuda_stream
    for seed, shape, iters, th in ((0, {}, 10, 15), (3, dict(n=12, wI=128, hI=96, gw=120, gh=90), 7, 4)):
        images, p2x, p2y, gw, gh = problem(seed, **shape)
        n, hI, wI = images.shape
        pc = np.zeros(gw * gh, np.float32)
        vf = np.ones(hI * wI, np.float32)
        want_er = []
        for it in range(iters):
            oth2 = 10000 * 10000 if it < iters // 2 else th * th
            pc, _, _, e1, r1 = oracle.vcal_plane_step(images, p2x, p2y, pc, vf, oth2)
            vf, _, _, e2, r2 = oracle.vcal_vignette_step(images, p2x, p2y, pc, vf, oth2)
            want_er.append((e1, r1, e2, r2))
        d_img, d_x, d_y = (torch.from_numpy(a).cuda() for a in (images, p2x, p2y))
        d_pc = torch.zeros(gw * gh, dtype=torch.float32, device="cuda")
        d_vf = torch.ones(hI * wI, dtype=torch.float32, device="cuda")
        er = ctx.vcal_solve(d_img, d_x, d_y, d_pc, d_vf, iters, th, st)
        assert bits_equal(d_pc.cpu().numpy(), pc) and bits_equal(d_vf.cpu().numpy(), vf), seed
        for it, (e1, r1, e2, r2) in enumerate(want_er):
            assert er[it, 1] == r1 and er[it, 3] == r2, (seed, it)
            assert abs(er[it, 0] - e1) <= 1e-9 * abs(e1) + 1e-9 and abs(er[it, 2] - e2) <= 1e-9 * abs(e2) + 1e-9, (seed, it)


def _smooth_cases():
    rng = np.random.default_rng(11)
    for w, h in ((48, 40), (7, 5), (1, 1), (2, 9), (257, 131)):
        v = rng.random(w * h).astype(np.float32)
        v[rng.random(w * h) < 0.3] = np.nan
        if w * h > 1000:  # a hole wider than the four passes can close, and an empty border band
            v[: w * 3] = np.nan
            v.reshape(h, w)[10:30, 5:25] = np.nan
        yield w, h, v


def test_oracle_smoothing_equals_reference_text(oracle):
    """The output smoothing :541-566 (four NaN-aware 3 x 3 mean passes): oracle == the reference's own lines."""
    from oracle import loader

    try:
        ref = loader.VcalRef()
    except OSError as e:
        pytest.skip(str(e))
    for w, h, v in _smooth_cases():
        for got, want in zip(oracle.vcal_smooth(v, w, h), ref.smooth(v, w, h)):
            assert bits_equal(got, want), (w, h)


@pytest.mark.gpu
def test_gpu_smoothing_is_bit_identical(oracle):
    import torch

    from mono_dataset_code_amd import capi

    ctx = capi.Context(0)
    st = torch.cuda.current_stream().cuda_stream
    for w, h, v in _smooth_cases():
        d_v = torch.from_numpy(v).cuda()
        tt, ct = ctx.vcal_smooth(d_v, w, h, st)
        torch.cuda.synchronize()
        want_tt, want_ct = oracle.vcal_smooth(v, w, h)
        assert bits_equal(tt.cpu().numpy(), want_tt) and bits_equal(ct.cpu().numpy(), want_ct), (w, h)
        assert bits_equal(d_v.cpu().numpy(), v)


def _coord_cases():
    rng = np.random.default_rng(1)
    gw, gh, wI, hI = 50, 40, 64, 48
    x = (rng.random(gw * gh) * 80 - 8).astype(np.float32)
    y = (rng.random(gw * gh) * 60 - 6).astype(np.float32)
    x[::17] = np.nan
    y[5::31] = np.inf
    x[3], y[3], x[4], x[7], x[8], x[9], y[10] = 1.5, 1.5, 1.4999, wI - 2.5, wI - 2.51, 3e9, -3e9
    return gw, gh, wI, hI, x, y


def test_oracle_coordinate_mask_equals_reference_text(oracle):
    """:345-357 (NaN coordinates for plane points outside the image), NaN / inf / out-of-int-range inputs included."""
    from oracle import loader

    try:
        ref = loader.VcalRef()
    except OSError as e:
        pytest.skip(str(e))
    gw, gh, wI, hI, x, y = _coord_cases()
    for got, want in zip(oracle.vcal_mask_coords(x, y, wI, hI), ref.mask_coords(x, y, gw, gh, wI, hI)):
        assert bits_equal(got, want)


@pytest.mark.gpu
def test_gpu_coordinate_mask_is_bit_identical(oracle):
    import torch

    from mono_dataset_code_amd import capi

    ctx = capi.Context(0)
    gw, gh, wI, hI, x, y = _coord_cases()
    d_x, d_y = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    ctx.vcal_mask_coords(d_x, d_y, wI, hI, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want_x, want_y = oracle.vcal_mask_coords(x, y, wI, hI)
    assert bits_equal(d_x.cpu().numpy(), want_x) and bits_equal(d_y.cpu().numpy(), want_y)


def _gradient_cases():
    rng = np.random.default_rng(3)
    for w, h, th in ((64, 48, 30), (9, 7, 10), (5, 5, 1), (4, 4, 1), (200, 90, 60), (333, 257, 25)):
        yy, xx = np.mgrid[0:h, 0:w]
        img = (100 + 30 * np.sin(0.2 * xx) * np.cos(0.15 * yy) + rng.normal(0, 8, (h, w))).astype(np.float32)
        img[rng.random((h, w)) < 0.02] += 200
        img[rng.random((h, w)) < 0.01] = np.nan
        yield w, h, th, img


def test_oracle_gradient_mask_equals_reference_text(oracle):
    """:293-301, the in-place, order-dependent gradient mask of a calibration image: oracle == the reference's own lines."""
    from oracle import loader

    try:
        ref = loader.VcalRef()
    except OSError as e:
        pytest.skip(str(e))
    for w, h, th, img in _gradient_cases():
        assert bits_equal(oracle.vcal_gradient_mask(img, th), ref.gradient_mask(img, th)), (w, h)


@pytest.mark.gpu
def test_gpu_gradient_mask_replays_the_sequential_loop(oracle):
    """The wavefront kernel (t = x + 5y) gives exactly the mask of the sequential raster-order loop, several images of a
    stack at once."""
    import torch

    from mono_dataset_code_amd import capi

    ctx = capi.Context(0)
    st = torch.cuda.current_stream().cuda_stream
    for w, h, th, img in _gradient_cases():
        stack = np.stack([img, img[::-1].copy(), np.roll(img, 3, axis=1)])
        d = torch.from_numpy(stack).cuda()
        ctx.vcal_gradient_mask(d, th, st)
        torch.cuda.synchronize()
        got = d.cpu().numpy()
        for k in range(3):
            assert bits_equal(got[k], oracle.vcal_gradient_mask(stack[k], th)), (w, h, k)


@pytest.mark.gpu
def test_gpu_image_scaling():
    """:286-291, image = meanExposure * image / exposure_time in float (0 counts as 1): the same two float operations as
    numpy's float32 arithmetic."""
    import torch

    from mono_dataset_code_amd import capi

    ctx = capi.Context(0)
    rng = np.random.default_rng(9)
    imgs = (rng.random((5, 37, 53)) * 255).astype(np.float32)
    imgs[1, 3, 4] = np.nan
    expo = np.array([0.0, 0.5, 13.25, 1e-3, 7.0], np.float32)
    mean = np.float32(3.7)
    want = np.stack([(mean * imgs[k]) / (np.float32(1) if expo[k] == 0 else expo[k]) for k in range(5)])
    d = torch.from_numpy(imgs).cuda()
    ctx.vcal_scale_images(d, mean, torch.from_numpy(expo).cuda(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bits_equal(d.cpu().numpy(), want)
