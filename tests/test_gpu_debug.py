"""The bounds-checking build of the kernels (-DMDC_DEBUG_BOUNDS=1: every LDS tap, staging chunk and gather index is
checked in the kernel and traps on a violation) run over random cameras and every tile shape: the host plan's
promises hold (no trap = no HIP error), and the results are those of the product build."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import bits_equal, test_frames as make_frames
from test_gpu_random import random_camera

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def debug_capi():
    from mono_dataset_code_amd import build

    path = build.build_debug()
    spec = importlib.util.spec_from_file_location("capi_debug", os.path.join(ROOT, "mono_dataset_code_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.LIB_HIP_PATH = path
    return m


@pytest.mark.parametrize("seed", range(6))
def test_bounds_checked_kernels_on_random_cameras(seed, tmp_path, debug_capi):
    import torch

    from mono_dataset_code_amd import capi, synth

    rng = np.random.default_rng(4000 + seed)
    lines = random_camera(rng)
    d = synth.write_sequence_calibration(str(tmp_path), lines)
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    W, H, w, h = fov.dims()
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), W, H)
    blob = capi.pack_tables(fov, photo)
    prod, dbg = capi.Context(0), debug_capi.Context(0)
    prod.import_tables(blob)
    dbg.import_tables(blob)
    frames = np.stack(make_frames(W, H, n_noise=3))
    n = len(frames)
    d_in = torch.from_numpy(frames).cuda()
    st = torch.cuda.current_stream().cuda_stream
    fin = torch.rand((n, W * H), device="cuda") * 255
    for cols, rows in ((64, 16), (64, 32), (64, 60), (64, 64), (128, 16), (128, 32), (0, 0)):
        for c, m in ((prod, capi), (dbg, debug_capi)):
            c.set_option(m.OPT_TILE_COLS, cols)
            c.set_option(m.OPT_TILE_ROWS, rows)
        for kernel in (capi.KERNEL_AUTO, capi.KERNEL_GATHER):
            prod.set_option(capi.OPT_KERNEL, kernel)
            dbg.set_option(debug_capi.OPT_KERNEL, kernel)
            for flags in (15, 8, 13):
                a = torch.full((n, w * h), -7.0, device="cuda")
                b = torch.full((n, w * h), -7.0, device="cuda")
                prod.process_batch(d_in.data_ptr(), a.data_ptr(), n, flags, st)
                dbg.process_batch(d_in.data_ptr(), b.data_ptr(), n, flags, st)
                torch.cuda.synchronize()  # a trapped kernel surfaces here as a HIP error
                assert bits_equal(a.cpu().numpy(), b.cpu().numpy()), (seed, cols, rows, kernel, flags)
            a = torch.full((n, w * h), -7.0, device="cuda")
            b = torch.full((n, w * h), -7.0, device="cuda")
            prod.undistort_batch_f32(fin.data_ptr(), a.data_ptr(), n, st)
            dbg.undistort_batch_f32(fin.data_ptr(), b.data_ptr(), n, st)
            torch.cuda.synchronize()
            assert bits_equal(a.cpu().numpy(), b.cpu().numpy()), (seed, cols, rows, kernel, "f32")
