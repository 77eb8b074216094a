"""Randomised parity: seeded random cameras (sizes, lens parameters, crop / full / explicit output
intrinsics) x random kernel options (tile shape, window buffers, XCD placement, frames per
workgroup, frame interleave) against the oracle, bit for bit.  Catches planning edge cases the
hand-picked cameras of conftest.py do not reach (windows that touch the frame border, tiles that
are entirely black, rows without taps, ragged last tiles)."""
import os

import numpy as np
import pytest

from conftest import bits_equal, test_frames as make_frames

pytestmark = pytest.mark.gpu


def random_camera(rng):
    W = int(rng.integers(5, 26)) * 16 if rng.random() < 0.8 else int(rng.integers(70, 400))  # some widths not % 16: gather path
    H = int(rng.integers(60, 300))
    w, h = int(rng.integers(24, 330)), int(rng.integers(17, 260))
    omega = float(rng.choice([0.0, 0.3, 0.7, 0.933271, 1.15]))
    fx, fy = rng.uniform(0.3, 0.6), rng.uniform(0.35, 0.7)
    cx, cy = rng.uniform(0.45, 0.55), rng.uniform(0.45, 0.55)
    mode = rng.choice(["crop", "full", "explicit"])
    l3 = mode if mode != "explicit" else "%.4f %.4f %.4f %.4f 0" % (rng.uniform(0.3, 0.8), rng.uniform(0.3, 0.9), rng.uniform(0.4, 0.6), rng.uniform(0.4, 0.6))
    return ("%.6f %.6f %.6f %.6f %.6f" % (fx, fy, cx, cy, omega), "%d %d" % (W, H), l3, "%d %d" % (w, h))


@pytest.mark.parametrize("seed", range(10))
def test_random_camera_random_options(seed, tmp_path, oracle):
    import torch

    from mono_dataset_code_amd import capi, synth

    rng = np.random.default_rng(1000 + seed)
    lines = random_camera(rng)
    d = synth.write_sequence_calibration(str(tmp_path), lines, vignette_bits=int(rng.choice([8, 16])))
    cam = oracle.parse_camera(os.path.join(d, "camera.txt"))
    assert cam["valid"], lines
    W, H, w, h = cam["in_w"], cam["in_h"], cam["out_w"], cam["out_h"]
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), W, H)
    assert fov.is_valid() and photo.valid() == 3
    rx, ry = fov.remap()
    ginv, vinv = photo.ginv(), photo.vignette()[1]
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    frames = np.stack(make_frames(W, H, n_noise=5))
    n = len(frames)
    d_in = torch.from_numpy(frames).cuda()
    st = torch.cuda.current_stream().cuda_stream
    want = {}
    for trial in range(6):
        rows = int(rng.choice([16, 32, 60, 64]))
        cols = int(rng.choice([64, 128])) if rows in (16, 32) else 64
        ctx.set_option(capi.OPT_TILE_COLS, cols)
        ctx.set_option(capi.OPT_TILE_ROWS, rows)
        ctx.set_option(capi.OPT_WINDOW_BUFFERS, int(rng.choice([0, 2, 3, 4])))
        ctx.set_option(capi.OPT_TILE_ORDER, int(rng.integers(0, 4)))
        ctx.set_option(capi.OPT_FRAMES_PER_BLOCK, int(rng.choice([0, 1, 2, 3, 5])))
        ctx.set_option(capi.OPT_FRAME_INTERLEAVE, int(rng.integers(0, 2)))
        g, v, o = (int(x) for x in rng.integers(0, 2, 3))
        flags = capi.RECTIFY | (capi.GAMMA * g) | (capi.VIGNETTE * v) | (capi.KILL_OVEREXPOSED * o)
        if (g, v, o) not in want:
            want[(g, v, o)] = np.stack([oracle.get_image(f, W, H, w, h, ginv, vinv, True, True, rx, ry, 1, g, v, o) for f in frames])
        levels = int(rng.integers(1, 5))
        d_out = torch.full((n, w * h), -7.0, dtype=torch.float32, device="cuda")
        lv = [torch.full((n * (w >> l) * (h >> l),), -7.0, dtype=torch.float32, device="cuda") for l in range(1, levels)]
        ctx.process_pyramid_batch(d_in.data_ptr(), d_out.data_ptr(), levels, [t.data_ptr() for t in lv], n, flags, st)
        torch.cuda.synchronize()
        info = ctx.info()
        tag = (seed, lines, trial, rows, cols, info.tiled, g, v, o, levels)
        assert bits_equal(d_out.cpu().numpy(), want[(g, v, o)]), tag
        for f in range(n):
            src, cw, ch = want[(g, v, o)][f], w, h
            for l in range(levels - 1):
                nxt = oracle.pyramid_level(src, cw, ch)
                assert bits_equal(lv[l].view(n, -1)[f].cpu().numpy(), nxt), tag + (f, l + 1)
                src, cw, ch = nxt, cw // 2, ch // 2
        # undistort<float> on the same frames (float source plan)
        if trial % 2 == 0:
            fin = np.stack([oracle.unmap(f, ginv, vinv, True, True, 1, 1, 1) for f in frames[:3]])
            d_f = torch.from_numpy(fin).cuda()
            d_u = torch.full((3, w * h), -7.0, dtype=torch.float32, device="cuda")
            ctx.undistort_batch_f32(d_f.data_ptr(), d_u.data_ptr(), 3, st)
            torch.cuda.synchronize()
            assert bits_equal(d_u.cpu().numpy(), np.stack([oracle.undistort(x, rx, ry, W) for x in fin])), tag
