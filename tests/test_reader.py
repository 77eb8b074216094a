"""class DatasetReader (include/mono_dataset_code/BenchmarkDatasetReader.h) on the GPU: the reference's reader
interface with the fused pass behind it.

  * getImage / getImages for every switch combination == the CPU oracle, bit for bit, from an images/ folder and
    from images.zip;
  * the playDataset-style driver compiled against OUR reader header (same source file as for the reference's
    reader) == the reference's own reader + classes end to end (playback_ref), byte for byte;
  * JPEG frames: identical to the oracle on the libjpeg-decoded bytes.
"""
import ctypes
import io
import itertools
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

from conftest import bits_equal
from test_reader_cpu import make_sequence, textured

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = {k: os.path.join(ROOT, "oracle", "_ref", k) for k in ("playback_ref", "playback_fast", "playback_batch")}
FLAGS = ["1111", "0111", "1000", "0000", "1110", "1010", "0001", "1100"]


def oracle_tables(d, oracle, w, h):
    from mono_dataset_code_amd import synth

    cam = oracle.parse_camera(os.path.join(d, "camera.txt"))
    t = oracle.fov_setup(cam)
    ginv, _ = oracle.photo_gamma(oracle.parse_pcalib(os.path.join(d, "pcalib.txt")))
    vinv = oracle.photo_vignette(synth.vignette_image(w, h, 16))[1]
    return cam, t, ginv, vinv


def frames_for(n, h, w):
    from mono_dataset_code_amd import synth

    fr = [textured(h, w, s) for s in range(n - 2)]
    fr.append(synth.smooth_frame(w, h, 0.7).reshape(h, w))  # saturated blobs -> NaN regions
    fr.append(np.full((h, w), 255, np.uint8))
    return fr


@pytest.mark.parametrize("zipped,fmt", [(False, "png"), (True, "png"), (True, "jpg"), (False, "pgm")])
def test_get_image_and_get_images_equal_the_oracle(tmp_path, oracle, zipped, fmt):
    from mono_dataset_code_amd import capi

    h, w = 256, 320
    frames = frames_for(9, h, w)
    names, blobs = make_sequence(str(tmp_path), frames, zipped, fmt)
    raw = frames if fmt != "jpg" else [np.asarray(Image.open(io.BytesIO(b))) for b in blobs]
    cam, t, ginv, vinv = oracle_tables(str(tmp_path), oracle, w, h)
    ow, oh = cam["out_w"], cam["out_h"]
    r = capi.DatasetReader(str(tmp_path))
    assert len(r) == 9
    for fl in itertools.product((0, 1), repeat=4):
        want = [oracle.get_image(f.reshape(-1), w, h, ow, oh, ginv, vinv, True, True, t["remap_x"], t["remap_y"], *fl) for f in raw]
        for i in (0, 8, 7, 3):
            img, ts, ex, idx = r.get_image(i, *fl)
            assert idx == i and abs(ts - (1000.0 + i / 20.0)) < 1e-9 and abs(ex - (1.0 + 0.01 * i)) < 1e-6
            assert img.shape == ((oh, ow) if fl[0] else (h, w))
            assert bits_equal(img, want[i]), (fl, i)
        out, ok, got = r.get_images(0, 9, *fl)
        assert got == 9 and ok.all()
        for i in range(9):
            assert bits_equal(out[i], want[i]), (fl, i, "getImages")
    out, ok, got = r.get_images(2, 5, 1, 1, 1, 1)  # a sub-range
    assert got == 5 and bits_equal(out[0], oracle.get_image(raw[2].reshape(-1), w, h, ow, oh, ginv, vinv, True, True, t["remap_x"], t["remap_y"], 1, 1, 1, 1))
    out, ok, got = r.get_images(5, 9, 1, 1, 1, 1)  # runs past the end: refused as a whole
    assert got == 0 and "outside the sequence" in r.last_error()
    r.close()


def test_get_images_longer_than_the_ring(tmp_path, oracle):
    """More frames than the page-locked ring holds (256 buffers, chunks of 32): chunk k+8 re-uses chunk k's buffers while the pool decodes."""
    from mono_dataset_code_amd import capi

    h, w = 64, 80
    frames = [textured(h, w, s) for s in range(300)]
    make_sequence(str(tmp_path), frames, True, "png")
    cam, t, ginv, vinv = oracle_tables(str(tmp_path), oracle, w, h)
    r = capi.DatasetReader(str(tmp_path))
    for threads in (0, 2):
        r.set_threads(threads)
        out, ok, got = r.get_images(0, 300, 1, 1, 1, 0)
        assert got == 300 and ok.all()
        for i in (0, 31, 32, 63, 64, 191, 192, 193, 250, 299):
            want = oracle.get_image(frames[i].reshape(-1), w, h, cam["out_w"], cam["out_h"], ginv, vinv, True, True, t["remap_x"], t["remap_y"], 1, 1, 1, 0)
            assert bits_equal(out[i], want), (threads, i)
    r.close()


def test_reader_shards_getimages_over_devices(tmp_path, monkeypatch):
    """MDC_DEVICES: getImages (and the results getImage makes ahead) dealt to several devices in chunks, each with its own
    context, decode ring and host thread -- here two lanes on the test box's one GPU ("0,0": a device listed twice takes the
    tables from the host objects; distinct devices get them in one RCCL broadcast).  Images, order, metadata and failures are
    those of the single-device reader, bit for bit, for JPEG streams (device Huffman stage), host-decoded JPEG, and PNG."""
    from mono_dataset_code_amd import capi

    h, w = 96, 160
    n = 300
    for fmt in ("jpg", "png"):
        d = os.path.join(str(tmp_path), fmt)
        os.makedirs(d)
        make_sequence(d, frames_for(n, h, w), True, fmt)
        monkeypatch.delenv("MDC_DEVICES", raising=False)
        one = capi.DatasetReader(d)
        monkeypatch.setenv("MDC_DEVICES", "0,0")
        two = capi.DatasetReader(d)
        monkeypatch.delenv("MDC_DEVICES", raising=False)
        assert len(one.device_stats()) == 1 and len(two.device_stats()) == 2
        for stage in ((2, 0) if fmt == "jpg" else (2,)):
            one.set_gpu_jpeg(stage)
            two.set_gpu_jpeg(stage)
            for first, count, fl in ((0, n, (1, 1, 1, 1)), (7, 129, (0, 1, 1, 0)), (250, 50, (1, 0, 0, 1)), (3, 40, (1, 1, 1, 1))):
                a, oka, na = one.get_images(first, count, *fl)
                b, okb, nb = two.get_images(first, count, *fl)
                assert na == nb == count and oka.all() and okb.all(), (fmt, stage, first, count)
                assert bits_equal(a, b), (fmt, stage, first, count)
        st = two.device_stats()
        assert all(dev == 0 and frames > 0 for dev, frames, _, _ in st), st  # both lanes produced frames
        assert sum(f for _, f, _, _ in st) == sum(c for _, c, _ in ((0, n, 0), (7, 129, 0), (250, 50, 0), (3, 40, 0))) * (2 if fmt == "jpg" else 1)
        # getImage in order: the results made ahead come out of the sharded getImages
        for i in list(range(0, 150)) + [5, 6, 7]:
            x = two.get_image(i, 1, 1, 1, 1)
            y = one.get_image(i, 1, 1, 1, 1)
            assert x is not None and x[1:] == y[1:] and bits_equal(x[0], y[0]), (fmt, i)
        one.close()
        two.close()


def test_reader_lanes_on_libmdc_multi_contexts(tmp_path, monkeypatch, capfd):
    """The reader's multi-device set-up as it runs with distinct devices -- libmdc_multi.so found next to libmdc_host.so and
    dlopen'ed, one communicator per device, rank 0's tables out in one RCCL broadcast, the lanes on libmdc_multi's contexts --
    executed on the test box's one GPU with a world of one (MDC_READER_FORCE_RCCL=1): same images as the plain reader."""
    from mono_dataset_code_amd import capi

    h, w = 96, 160
    d = str(tmp_path)
    make_sequence(d, frames_for(40, h, w), True, "jpg")
    plain = capi.DatasetReader(d)
    monkeypatch.setenv("MDC_DEVICES", "0")
    monkeypatch.setenv("MDC_READER_FORCE_RCCL", "1")
    capfd.readouterr()
    rccl = capi.DatasetReader(d)
    ctypes.CDLL(None).fflush(None)
    out = capfd.readouterr().out
    monkeypatch.delenv("MDC_DEVICES")
    monkeypatch.delenv("MDC_READER_FORCE_RCCL")
    assert "calibration tables broadcast over RCCL" in out, out[-800:]
    a, oka, na = plain.get_images(0, 40, 1, 1, 1, 1)
    b, okb, nb = rccl.get_images(0, 40, 1, 1, 1, 1)
    assert na == nb == 40 and oka.all() and okb.all() and bits_equal(a, b)
    x, y = plain.get_image(7, 0, 1, 1, 0), rccl.get_image(7, 0, 1, 1, 0)
    assert x[1:] == y[1:] and bits_equal(x[0], y[0])
    assert rccl.device_stats()[0][0] == 0
    plain.close()
    rccl.close()


def test_reader_with_the_image_pool_switched_off(tmp_path):
    """MDC_IMAGE_POOL=0 (ExposureImage::image = new float[], the reference's own allocation): getImage / getImages give the same
    images; the GPU reaches them through staging copies instead of writing page-locked slabs in place."""
    import subprocess
    import sys

    from mono_dataset_code_amd import capi

    h, w = 96, 160
    d = str(tmp_path)
    make_sequence(d, frames_for(70, h, w), True, "jpg")
    r = capi.DatasetReader(d)
    want, ok, n = r.get_images(0, 70, 1, 1, 1, 1)
    one = r.get_image(11, 1, 1, 0, 1)
    r.close()
    assert n == 70 and ok.all()
    np.save(os.path.join(d, "want.npy"), want)
    np.save(os.path.join(d, "one.npy"), one[0])
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
from mono_dataset_code_amd import capi
r = capi.DatasetReader(%r)
got, ok, n = r.get_images(0, 70, 1, 1, 1, 1)
one = r.get_image(11, 1, 1, 0, 1)
assert n == 70 and ok.all()
want = np.load(%r); w1 = np.load(%r)
eq = lambda a, b: np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32))
assert eq(got, want) and eq(one[0], w1)
print("POOL_OFF_OK")
""" % (ROOT, d, os.path.join(d, "want.npy"), os.path.join(d, "one.npy"))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MDC_IMAGE_POOL="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0 and "POOL_OFF_OK" in p.stdout, p.stderr[-2000:]


def parse(path):
    raw = open(path, "rb").read()
    pos, recs = 0, []
    while pos < len(raw):
        w, h, idx, a = np.frombuffer(raw, np.int32, 4, pos)
        ts = np.frombuffer(raw, np.float64, 1, pos + 16)[0]
        ex = np.frombuffer(raw, np.float32, 1, pos + 24)[0]
        recs.append((int(w), int(h), int(idx), int(a), float(ts), float(ex), np.frombuffer(raw, np.float32, w * h, pos + 28)))
        pos += 28 + 4 * w * h
    return recs


@pytest.mark.parametrize("zipped", [False, True])
@pytest.mark.parametrize("which", ["playback_fast", "playback_batch"])
def test_playdataset_driver_on_our_reader_equals_the_reference_reader(tmp_path, which, zipped):
    """tests/dropin/playback_headless.cpp is ONE source file: compiled against the reference's BenchmarkDatasetReader.h +
    classes (playback_ref, CPU) and against this repo's reader header (playback_fast, GPU); playback_batch drives
    getImages().  Same bytes out, from an images/ folder and from images.zip."""
    for b in ("playback_ref", which):
        if not os.path.exists(BIN[b]):
            pytest.skip("%s not built" % b)
    h, w = 256, 320
    frames = frames_for(5, h, w)
    d = str(tmp_path / "seq")
    os.makedirs(d)
    # zipped: the reference's unmodified reader goes through its zip_fopen / zip_fread / cv::imdecode path
    # (src/BenchmarkDatasetReader.h:256-274; oracle/shim_zip.cpp + libpng, independent of the product's zip reader and decoders)
    make_sequence(d, frames, zipped, "png")
    outs = {}
    for b in ("playback_ref", which):
        out = str(tmp_path / (b + ".bin"))
        r = subprocess.run([BIN[b], d, out] + FLAGS, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-3000:]
        outs[b] = parse(out)
    assert len(outs["playback_ref"]) == len(outs[which]) == len(FLAGS) * 5
    for x, y in zip(outs["playback_ref"], outs[which]):
        assert x[:6] == y[:6]
        assert bits_equal(x[6], y[6]), x[:4]


@pytest.mark.parametrize("size", [(64, 64), (100, 130), (17, 23), (250, 322), (256, 320)])
def test_gpu_jpeg_stage_equals_the_host_decoder(size):
    """Row f2's GPU JPEG stage: host Huffman decoding into coefficient records (mdch_decode_jpeg_record) + dequantisation and
    libjpeg's islow inverse DCT on the device (mdc_jpeg_idct_batch_device) == the host decoder (itself pinned to libjpeg-turbo
    in test_reader_cpu.py), byte for byte: gray and YCbCr with every chroma subsampling, baseline and progressive, restart
    markers, sizes that are not whole blocks."""
    import torch

    from mono_dataset_code_amd import capi

    h, w = size
    img = textured(h, w, 4)
    rgb = np.stack([img, np.roll(img, 3, 1), 255 - img], -1)
    files = []
    for kw in ({"quality": 30}, {"quality": 92}, {"quality": 100}, {"quality": 80, "restart_marker_blocks": 7}, {"quality": 85, "progressive": True}):
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", **kw)
        files.append(b.getvalue())
    for sub in (0, 1, 2):
        for prog in (False, True):
            b = io.BytesIO()
            Image.fromarray(rgb).save(b, "JPEG", quality=88, subsampling=sub, progressive=prog)
            files.append(b.getvalue())
    rec_bytes, pitch, rows = capi.jpeg_record_bytes(w, h)
    recs = np.zeros((len(files), rec_bytes), np.uint8)
    for i, data in enumerate(files):
        dims = capi.decode_jpeg_record(data, recs[i], pitch)
        assert dims[:3] == (w, h, pitch) and dims[3] <= rows
    ctx = capi.Context(0)
    d_rec = torch.from_numpy(recs).cuda()
    d_frames = torch.full((len(files), h * w), 77, dtype=torch.uint8, device="cuda")
    ctx.jpeg_idct_batch(d_rec.data_ptr(), rec_bytes, d_frames.data_ptr(), w, h, pitch, rows, len(files), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_frames.cpu().numpy().reshape(len(files), h, w)
    for i, data in enumerate(files):
        assert np.array_equal(got[i], capi.decode_gray8(data)), i


@pytest.mark.parametrize("size", [(64, 64), (100, 130), (17, 23), (250, 322), (256, 320), (1024, 1280)])
def test_gpu_huffman_stage_equals_the_host_decoder(size):
    """The Huffman decoding itself on the device (mdch_jpeg_stream: markers parsed, tables built, byte stuffing removed on the
    host; mdc_jpeg_huffman_batch_device: 1024 threads per stream, subsequence states relaxed to the sequential decoder's):
    the coefficient RECORD equals the host decoder's record byte for byte -- quantisation table, every coefficient of every
    block, DC prediction -- for textures, noise (long codes, few zero runs), flat images (blocks of one EOB), qualities 5..100,
    optimised Huffman tables, sizes that are not whole blocks; files with restart intervals (each interval decoded from its
    exact entry state) and YCbCr files with the three components interleaved (4:4:4, 4:2:2, 4:2:0: the luma plane is what
    cv::imread(..., GRAYSCALE) keeps), all kinds mixed in one batch; then the device inverse DCT gives the host decoder's
    pixels.  Files the device does not take (progressive) are refused by mdch_jpeg_stream."""
    import torch

    from mono_dataset_code_amd import capi

    h, w = size
    rng = np.random.default_rng(h * 7 + w)
    imgs = [textured(h, w, 4), rng.integers(0, 256, (h, w), dtype=np.uint8), np.full((h, w), 131, np.uint8),
            (np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8)]
    files = []
    for k, img in enumerate(imgs):
        rgb = np.stack([img, np.roll(img, 3, 1), 255 - img], -1)
        for kw in ({"quality": 5}, {"quality": 50}, {"quality": 92}, {"quality": 100}, {"quality": 75, "optimize": True},
                   {"quality": 92, "restart_marker_blocks": 7}, {"quality": 60, "restart_marker_rows": 1}, {"quality": 95, "restart_marker_blocks": 1},
                   {"quality": 92, "subsampling": 0, "rgb": 1}, {"quality": 70, "subsampling": 1, "rgb": 1}, {"quality": 100, "subsampling": 2, "rgb": 1},
                   {"quality": 85, "subsampling": 2, "rgb": 1, "restart_marker_rows": 1}, {"quality": 30, "subsampling": 1, "rgb": 1, "optimize": True,
                                                                                         "restart_marker_blocks": 3}):
            if h * w > 500000 and kw["quality"] not in (92, 100):
                continue
            kw = dict(kw)
            src = rgb if kw.pop("rgb", 0) else img
            b = io.BytesIO()
            Image.fromarray(src).save(b, "JPEG", **kw)
            files.append(b.getvalue())
    rec_bytes, pitch, rows = capi.jpeg_record_bytes(w, h)
    n = len(files)
    want = np.zeros((n, rec_bytes), np.uint8)
    cap = (2 * capi.JPEG_STREAM_HEADER_BYTES + 4 * ((w + 7) // 8) * ((h + 7) // 8) + max(len(f) for f in files) + 64 + 15) & ~15
    streams = np.zeros((n, cap), np.uint8)
    for i, data in enumerate(files):
        dims = capi.decode_jpeg_record(data, want[i], pitch)
        assert dims[:3] == (w, h, pitch)
        used, sw, sh = capi.jpeg_stream(data, streams[i])
        assert (sw, sh) == (w, h) and capi.JPEG_STREAM_HEADER_BYTES + 17 <= used <= cap
    ctx = capi.Context(0)
    st = torch.cuda.current_stream().cuda_stream
    d_streams = torch.from_numpy(streams).cuda()
    d_rec = torch.full((n, rec_bytes), 0x5A, dtype=torch.uint8, device="cuda")
    d_status = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    ctx.jpeg_huffman_batch(d_streams.data_ptr(), cap, d_rec.data_ptr(), rec_bytes, w, h, pitch, rows, n, d_status.data_ptr(), st)
    d_frames = torch.full((n, h * w), 77, dtype=torch.uint8, device="cuda")
    ctx.jpeg_idct_batch(d_rec.data_ptr(), rec_bytes, d_frames.data_ptr(), w, h, pitch, rows, n, st)
    torch.cuda.synchronize()
    assert d_status.cpu().numpy().tolist() == [0] * n
    got = d_rec.cpu().numpy()
    bw, bh = (w + 7) // 8, (h + 7) // 8
    for i in range(n):
        assert np.array_equal(got[i, :128], want[i, :128]), (i, "quantisation table")
        g = got[i, 128:].view(np.int16).reshape(rows, pitch, 64)[:bh, :bw]
        e = want[i, 128:].view(np.int16).reshape(rows, pitch, 64)[:bh, :bw]
        assert np.array_equal(g, e), (i, int((g != e).any(-1).sum()), "blocks differ", np.argwhere((g != e).any(-1))[:4].tolist())
        assert np.array_equal(d_frames[i].cpu().numpy().reshape(h, w), capi.decode_gray8(files[i])), i
    # what the device decoder does not take is refused on the host
    rgb = np.stack([imgs[0], np.roll(imgs[0], 3, 1), 255 - imgs[0]], -1)
    for img, kw in ((rgb, {"quality": 88, "progressive": True}), (imgs[0], {"quality": 85, "progressive": True})):
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", **kw)
        with pytest.raises(ValueError):
            capi.jpeg_stream(b.getvalue(), streams[0].copy())
    # a stream whose bits were damaged is reported, not decoded into something
    bad = streams[:2].copy()  # (the first two files: one component, no restart markers -- the stream body follows the header directly)
    bad[0, capi.JPEG_STREAM_HEADER_BYTES + 40: capi.JPEG_STREAM_HEADER_BYTES + 60] ^= 0xFF
    bad[1, 4] ^= 1  # header: another width
    d_bad = torch.from_numpy(bad).cuda()
    ctx.jpeg_huffman_batch(d_bad.data_ptr(), cap, d_rec.data_ptr(), rec_bytes, w, h, pitch, rows, 2, d_status.data_ptr(), st)
    torch.cuda.synchronize()
    s2 = d_status.cpu().numpy()[:2].tolist()
    assert s2[1] == 2 and s2[0] in (0, 1)  # (flipped bits may still parse as SOME valid stream of the right length: then 0)
    # tables no JPEG file produces -- every entry a code of length 0 -- must end in a status, not in an endless loop
    hostile = streams[:1].copy()
    huff = 2048 * 4 + 32 * 32 * 4  # sizeof(mdc_jpeg_huff): t1, t2
    assert capi.JPEG_STREAM_HEADER_BYTES == 160 + 2 * huff
    hostile[0, 160:160 + 8192].view(np.uint32)[:] = 31 | 5 << 16           # dc.t1: every window -> subtable 5 ...
    hostile[0, 160 + 8192:160 + huff].view(np.uint32)[:] = 0               # ... which holds no code
    hostile[0, 160 + huff:160 + huff + 8192].view(np.uint32)[:] = 17 | 3 << 5  # ac.t1: codes of "length 17"
    d_h = torch.from_numpy(hostile).cuda()
    ctx.jpeg_huffman_batch(d_h.data_ptr(), cap, d_rec.data_ptr(), rec_bytes, w, h, pitch, rows, 1, d_status.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(d_status[0]) == 1


def huffman_batches_equal_the_host_decoder(capi_dev, size, batch_sizes):
    """Records of mdc_jpeg_huffman_batch_device (through `capi_dev`: the product library or a variant build) against the host
    decoder's, for batches of one-component and interleaved YCbCr files, with and without restart intervals (long ones: cut into
    parts that relax inside a wave, several workgroups per frame in small batches), every batch twice on the same buffers."""
    import torch

    from mono_dataset_code_amd import capi

    h, w = size
    rng = np.random.default_rng(h + w)
    files = []
    for k in range(8):  # 0-3 one component, 4 / 5 YCbCr 4:4:4 / 4:2:0, 6 / 7 a restart interval per MCU row (one component / 4:2:0)
        img = textured(h, w, k) if k % 3 != 2 else rng.integers(0, 256, (h, w), dtype=np.uint8)
        if k in (4, 5, 7):
            img = np.stack([img, np.roll(img, 3, 1), 255 - img], -1)
        kw = {"subsampling": 2} if k in (5, 7) else {}
        if k >= 6:
            kw["restart_marker_rows"] = 1
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=(35, 90, 97, 75, 88, 60, 90, 80)[k], **kw)
        files.append(b.getvalue())
    rec_bytes, pitch, rows = capi.jpeg_record_bytes(w, h)
    cap = (2 * capi.JPEG_STREAM_HEADER_BYTES + 4 * ((w + 7) // 8) * ((h + 7) // 8) + max(len(f) for f in files) + 64 + 15) & ~15
    want = np.zeros((len(files), rec_bytes), np.uint8)
    one = np.zeros((len(files), cap), np.uint8)
    for i, data in enumerate(files):
        capi.decode_jpeg_record(data, want[i], pitch)
        capi.jpeg_stream(data, one[i])
    used = 128 + rows * pitch * 128
    bw, bh = (w + 7) // 8, (h + 7) // 8
    d_want = torch.from_numpy(want).cuda()[:, 128:used].view(torch.int16).reshape(len(files), rows, pitch, 64)[:, :bh, :bw]
    d_quant = torch.from_numpy(want).cuda()[:, :128]
    d_one = torch.from_numpy(one).cuda()
    ctx = capi_dev.Context(0)
    st = torch.cuda.current_stream().cuda_stream
    nmax = max(batch_sizes)
    d_rec = torch.empty((nmax, rec_bytes), dtype=torch.uint8, device="cuda")
    d_status = torch.empty((nmax,), dtype=torch.int32, device="cuda")
    for n in batch_sizes:
        idx = (torch.arange(n, device="cuda") * 5 + n) % len(files)  # another order of the files for every batch size
        d_streams = d_one[idx].contiguous()
        for call in range(2):
            d_rec.fill_(0x5A)
            d_status.fill_(-1)
            ctx.jpeg_huffman_batch(d_streams.data_ptr(), cap, d_rec.data_ptr(), rec_bytes, w, h, pitch, rows, n, d_status.data_ptr(), st)
            torch.cuda.synchronize()
            assert bool((d_status[:n] == 0).all()), (n, call, d_status[:n].cpu().tolist())
            got = d_rec[:n, 128:used].view(torch.int16).reshape(n, rows, pitch, 64)[:, :bh, :bw]
            assert torch.equal(got, d_want[idx]), (n, call, torch.nonzero((got != d_want[idx]).any(-1))[:4].cpu().tolist())
            assert torch.equal(d_rec[:n, :128], d_quant[idx]), (n, call)


@pytest.mark.parametrize("size", [(1024, 1280), (480, 640), (40, 56)])
def test_gpu_huffman_every_number_of_workgroups_per_frame(size):
    """Small batches spread a frame's stream over 8 / 4 / 2 workgroups that hand provisional, then final exit states on
    (jpeg_huffman_split_kernel); large ones take one workgroup per frame: the record is the host decoder's whatever the
    batch size -- 1, 3, 16 (8 per frame), 17, 33, 64 (4), 65, 100 (2), 130 (1) -- for one-component and interleaved YCbCr files in
    one batch, and a second call on the same buffers gives the same bytes (the segment states are re-initialised per call)."""
    from mono_dataset_code_amd import capi

    huffman_batches_equal_the_host_decoder(capi, size, (1, 3, 16, 17, 33, 64, 65, 100, 130))


@pytest.mark.parametrize("size", [(1024, 1280), (100, 130)])
def test_gpu_huffman_with_wrong_provisional_states(size):
    """Fault injection (mono_dataset_code_amd/build.py:build_fault_injection, -DMDC_EXP_HUFF_BAD_PROVISIONAL): every segment
    publishes a WRONG provisional exit state, so every right neighbour relaxes from it, finds the final state different and
    relaxes a third time -- the path a product run takes only when a guessed entry state never resynchronises (one-component
    files: practically never; colour files: the position in the MCU is part of the state and mostly guessed wrong).  Same
    records as the host decoder."""
    import importlib.util

    from mono_dataset_code_amd import build

    path = build.build_fault_injection()
    spec = importlib.util.spec_from_file_location("capi_badprov", os.path.join(ROOT, "mono_dataset_code_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.LIB_HIP_PATH = path
    assert "MDC_EXP_HUFF_BAD_PROVISIONAL" in m.build_flags()
    huffman_batches_equal_the_host_decoder(m, size, (2, 16, 20, 64, 100))


def test_gpu_huffman_stage_random_files():
    """300 random files -- sizes 1x1 .. 97x131, qualities 1..100, optimised tables or the standard ones, one component or YCbCr
    (4:4:4 / 4:2:2 / 4:2:0), with or without restart intervals of 1..11 MCUs, content from flat
    through gradients and sparse dots to noise (the long codes, ZRL runs, 63-coefficient blocks and one-symbol blocks the
    tables can produce; streams with many FF bytes to unstuff) -- each through the device Huffman decoder and the host
    decoder: identical records, file by file (streams padded to one stride, one launch per size)."""
    import torch

    from mono_dataset_code_amd import capi

    rng = np.random.default_rng(20260926)
    ctx = capi.Context(0)
    st = torch.cuda.current_stream().cuda_stream
    checked = 0
    for trial in range(32):
        h, w = int(rng.integers(1, 98)), int(rng.integers(1, 132))
        if trial >= 30:  # ... and the sizes sequences really have: every flavour once more at 640 x 480 and 1280 x 1024
            h, w = ((480, 640), (1024, 1280))[trial - 30]
        files = []
        for k in range(10 if trial < 30 else 4):
            kind = int(rng.integers(0, 6))
            if kind == 0:
                img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
            elif kind == 1:
                img = rng.integers(0, 256, (h, w), dtype=np.uint8)
            elif kind == 2:
                img = ((np.add.outer(np.arange(h) * int(rng.integers(1, 9)), np.arange(w) * int(rng.integers(1, 9)))) % 256).astype(np.uint8)
            elif kind == 3:
                img = np.zeros((h, w), np.uint8)
                img[rng.integers(0, h, 5), rng.integers(0, w, 5)] = 255
            elif kind == 4:
                img = (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)
            else:
                img = np.clip(rng.normal(128, float(rng.integers(1, 90)), (h, w)), 0, 255).astype(np.uint8)
            kw = {"quality": int(rng.integers(1, 101)), "optimize": bool(rng.integers(0, 2))}
            flavour = int(rng.integers(0, 4)) if trial < 30 else k  # plain / restart intervals / YCbCr / both
            if flavour & 1:
                kw["restart_marker_blocks"] = int(rng.integers(1, 12))
            if flavour & 2:
                img = np.stack([img, np.roll(img, 1, 0), 255 - img], -1)
                kw["subsampling"] = int(rng.integers(0, 3))
            b = io.BytesIO()
            Image.fromarray(img).save(b, "JPEG", **kw)
            files.append(b.getvalue())
        n = len(files)
        rec_bytes, pitch, rows = capi.jpeg_record_bytes(w, h)
        want = np.zeros((n, rec_bytes), np.uint8)
        cap = (2 * capi.JPEG_STREAM_HEADER_BYTES + 4 * ((w + 7) // 8) * ((h + 7) // 8) + max(len(f) for f in files) + 64 + 15) & ~15
        streams = np.zeros((n, cap), np.uint8)
        for i, data in enumerate(files):
            capi.decode_jpeg_record(data, want[i], pitch)
            capi.jpeg_stream(data, streams[i])
        d_streams = torch.from_numpy(streams).cuda()
        d_rec = torch.full((n, rec_bytes), 0x5A, dtype=torch.uint8, device="cuda")
        d_status = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        ctx.jpeg_huffman_batch(d_streams.data_ptr(), cap, d_rec.data_ptr(), rec_bytes, w, h, pitch, rows, n, d_status.data_ptr(), st)
        torch.cuda.synchronize()
        assert d_status.cpu().numpy().tolist() == [0] * n, (trial, h, w)
        got = d_rec.cpu().numpy()
        bw, bh = (w + 7) // 8, (h + 7) // 8
        for i in range(n):
            assert np.array_equal(got[i, :128], want[i, :128]), (trial, i, h, w, "quantisation table")
            g = got[i, 128:].view(np.int16).reshape(rows, pitch, 64)[:bh, :bw]
            e = want[i, 128:].view(np.int16).reshape(rows, pitch, 64)[:bh, :bw]
            assert np.array_equal(g, e), (trial, i, h, w, len(files[i]))
            checked += 1
    assert checked == 308


def test_get_image_results_made_ahead_on_jpeg_sequences(tmp_path):
    """getImage on a JPEG sequence read in order: from the third consecutive id on the next frames come out of one pass of the
    getImages pipeline (setResultLookahead).  Every image -- made ahead or not, across batch borders, after a change of the
    switches, after a jump backwards, at the end of the sequence -- equals the one the reader gives with the lookahead off,
    metadata included."""
    from mono_dataset_code_amd import capi

    h, w = 96, 160
    frames = frames_for(23, h, w)
    make_sequence(str(tmp_path), frames, True, "jpg")
    ref = capi.DatasetReader(str(tmp_path))
    ref.set_lookahead(0)
    r = capi.DatasetReader(str(tmp_path))
    r.set_lookahead(5)
    walk = [(i, (1, 1, 1, 1)) for i in range(0, 14)] + [(i, (1, 1, 0, 1)) for i in range(14, 19)] + [(3, (1, 1, 0, 1)), (4, (1, 1, 0, 1))] + \
           [(i, (0, 1, 1, 0)) for i in range(5, 23)] + [(22, (0, 1, 1, 0)), (0, (1, 1, 1, 1))]
    for i, fl in walk:
        a = r.get_image(i, *fl)
        b = ref.get_image(i, *fl)
        assert a is not None and b is not None, (i, fl)
        assert a[1:] == b[1:] and a[0].shape == b[0].shape, (i, fl)
        assert bits_equal(a[0], b[0]), (i, fl)
    r.close()
    ref.close()
    # the default: batches of 64, 128, 256 while the caller keeps reading in order (here 2 + 64 + 128 + the rest), back to 64 after a jump
    d2 = os.path.join(str(tmp_path), "long")
    os.makedirs(d2)
    make_sequence(d2, frames_for(230, h, w), True, "jpg")
    ref = capi.DatasetReader(d2)
    ref.set_lookahead(0)
    r = capi.DatasetReader(d2)
    for i in list(range(0, 230)) + list(range(40, 50)):
        a = r.get_image(i, 1, 1, 1, 1)
        b = ref.get_image(i, 1, 1, 1, 1)
        assert a is not None and a[1:] == b[1:] and bits_equal(a[0], b[0]), i
    r.close()
    ref.close()


def test_reader_gpu_jpeg_stages_agree_on_damaged_and_mixed_files(tmp_path, capfd):
    """A folder whose JPEGs are not all what the device Huffman decoder takes or can decode: gray baseline files, a colour file,
    a progressive one, one with restart markers (stage 2 refuses them on the host: they take the record path), a file whose
    entropy-coded bytes were damaged and a truncated one (the device reports them or decodes the same symbols as the host).
    getImages must give the same images -- and the same failures -- in stage 2, stage 1 and on the host."""
    from mono_dataset_code_amd import capi, synth

    h, w = 96, 160
    base = [textured(h, w, s) for s in range(9)]
    d = str(tmp_path)
    cam = ("0.349153 0.436593 0.493140 0.499021 0.933271", "%d %d" % (w, h), "crop", "%d %d" % (w * 3 // 5, h * 9 // 16))
    synth.write_sequence_calibration(d, cam, vignette_bits=16, n_times=len(base))
    os.makedirs(os.path.join(d, "images"))

    def jpeg(img, **kw):
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", **kw)
        return bytearray(b.getvalue())

    blobs = [jpeg(base[0], quality=90), jpeg(np.stack([base[1], base[1] // 2, 255 - base[1]], -1), quality=90),
             jpeg(base[2], quality=85, progressive=True), jpeg(base[3], quality=80, restart_marker_blocks=5), jpeg(base[4], quality=95),
             jpeg(base[5], quality=90), jpeg(base[6], quality=90), jpeg(base[7], quality=30, optimize=True), jpeg(base[8], quality=100)]
    sos = bytes(blobs[5]).index(b"\xff\xda")
    for k in range(sos + 400, sos + 420):  # damaged scan data (no FF bytes introduced: still one scan)
        blobs[5][k] = (blobs[5][k] ^ 0x55) if (blobs[5][k] ^ 0x55) != 0xff else 0x54
    blobs[6] = blobs[6][: sos + (len(blobs[6]) - sos) // 2] + b"\xff\xd9"  # truncated scan
    for i, b in enumerate(blobs):
        open(os.path.join(d, "images", "%05d.jpg" % i), "wb").write(bytes(b))
    results = {}
    for stage in (2, 1, 0):
        r = capi.DatasetReader(d)
        r.set_gpu_jpeg(stage)
        imgs, ok, n = r.get_images(0, len(blobs), 1, 1, 1, 1)
        results[stage] = (imgs, ok.copy(), n)
        r.close()
    ok0 = results[0][1]
    assert ok0[[0, 1, 2, 3, 4, 7, 8]].all()
    # getImage in order over the same folder: results made ahead (stage 2, lookahead) or not, the caller sees the same images, the
    # same failures and the same ERROR lines -- a frame that failed in the batch made ahead is reported when it is asked for, once
    logs = {}
    for look in (4, 0):
        r = capi.DatasetReader(d)
        r.set_lookahead(look)
        ctypes.CDLL(None).fflush(None)  # (libc's buffer: the C++ side printf's)
        capfd.readouterr()
        got = [r.get_image(i, 1, 1, 1, 1) for i in range(len(blobs))]
        r.close()
        ctypes.CDLL(None).fflush(None)
        logs[look] = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("ERROR")]
        for i in range(len(blobs)):
            assert (got[i] is not None) == bool(ok0[i]), (look, i)
            if ok0[i]:
                assert bits_equal(got[i][0], results[0][0][i]), (look, i)
    assert logs[4] == logs[0] and len(logs[0]) == int((~ok0).sum())
    for stage in (2, 1):
        assert results[stage][2] == results[0][2] and (results[stage][1] == ok0).all(), (stage, results[stage][1], ok0)
        for i in range(len(blobs)):
            if ok0[i]:
                assert bits_equal(results[stage][0][i], results[0][0][i]), (stage, i)
    # getImagesDevice over the same folder: the same images at the same positions, the same failures (valid[i] = 0, the position
    # untouched), in every stage -- damaged streams take the host decoder and go up as one frame, undecodable ones are refused
    import torch

    npo = results[0][0].shape[1]
    for stage in (2, 1, 0):
        r = capi.DatasetReader(d)
        r.set_gpu_jpeg(stage)
        d_base = torch.full((len(blobs), npo), -9.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        valid, n = r.get_images_device(0, len(blobs), 1, 1, 1, 1, capi.DeviceOutputs.make(d_base.data_ptr()))
        r.close()
        got = d_base.cpu().numpy()
        assert n == results[0][2] and (valid == ok0).all(), (stage, valid, ok0)
        for i in range(len(blobs)):
            if ok0[i]:
                assert bits_equal(got[i], results[0][0][i]), (stage, i)
            else:
                assert (got[i] == -9.0).all(), (stage, i)


def test_reader_gpu_jpeg_on_and_off_give_the_same_images(tmp_path):
    """getImages over a zipped JPEG sequence with the GPU JPEG stage (default) and without: identical ExposureImages."""
    from mono_dataset_code_amd import capi

    h, w = 256, 320
    frames = frames_for(40, h, w)  # longer than one chunk of 32
    names, blobs = make_sequence(str(tmp_path), frames, True, "jpg")
    r = capi.DatasetReader(str(tmp_path))
    on, ok_on, n_on = r.get_images(0, 40, 1, 1, 1, 1)  # default: Huffman decoding on the device too
    r.set_gpu_jpeg(1)
    mid, ok_mid, n_mid = r.get_images(0, 40, 1, 1, 1, 1)  # host Huffman, device inverse DCT
    r.set_gpu_jpeg(False)
    off, ok_off, n_off = r.get_images(0, 40, 1, 1, 1, 1)
    assert n_on == n_mid == n_off == 40 and ok_on.all() and ok_mid.all() and ok_off.all()
    for i in range(40):
        assert bits_equal(on[i], off[i]), i
        assert bits_equal(mid[i], off[i]), i
    # the C entry point itself, results into ordinary (pageable) arrays -- the staged way out -- and into page-locked ones, for
    # the photometric-only flags too; 70 streams = more than one 64-frame chunk: decode and output streams hand over
    d = str(tmp_path)
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), w, h)
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    cap = (capi.JPEG_STREAM_HEADER_BYTES + max(len(b) for b in blobs) + 64 + 15) & ~15
    idx = [i % 40 for i in range(70)]
    pin = capi.PinnedArray((70, cap), np.uint8)
    sizes = [capi.jpeg_stream(blobs[i], pin.array[k])[0] for k, i in enumerate(idx)]
    streams = [pin.array[k] for k in range(70)]
    loose = [np.full(on.shape[1], -7.0, np.float32) for _ in idx]
    assert ctx.process_jpeg_streams_host(streams, sizes, loose, 15) == [0] * 70
    pout = capi.PinnedArray((70, on.shape[1]), np.float32)
    assert ctx.process_jpeg_streams_host(streams, sizes, [pout.array[k] for k in range(70)], 15) == [0] * 70
    for k, i in enumerate(idx):
        assert bits_equal(loose[k], off[i]), (k, "pageable")
        assert bits_equal(pout.array[k], off[i]), (k, "page-locked")
    # streams back to back at one pitch in a block that ENDS with the last, shortest stream: the chunk goes up as one strided copy
    # of the first n-1 rows and the last stream by itself (nothing is read past the caller's block)
    order = sorted(range(40), key=lambda i: -len(blobs[i]))[:9]
    pitch = (capi.JPEG_STREAM_HEADER_BYTES + len(blobs[order[0]]) + 64 + 15) & ~15
    probe = np.zeros(pitch, np.uint8)
    last_size = capi.jpeg_stream(blobs[order[-1]], probe)[0]
    tight = capi.PinnedArray(((len(order) - 1) * pitch + last_size,), np.uint8)
    tsizes = []
    for k, i in enumerate(order):  # (mdch_jpeg_stream wants room for the file's worst case: written aside, then placed)
        tsizes.append(capi.jpeg_stream(blobs[i], probe)[0])
        tight.array[k * pitch: k * pitch + tsizes[-1]] = probe[: tsizes[-1]]
    assert tsizes[-1] == last_size and last_size < max(tsizes)
    tout = [np.full(on.shape[1], -7.0, np.float32) for _ in order]
    assert ctx.process_jpeg_streams_host([tight.array[k * pitch: k * pitch + tsizes[k]] for k in range(len(order))], tsizes, tout, 15) == [0] * len(order)
    for k, i in enumerate(order):
        assert bits_equal(tout[k], off[i]), (k, "tight block")
    raw_out = [np.full(w * h, -7.0, np.float32) for _ in range(3)]
    assert ctx.process_jpeg_streams_host(streams[:3], sizes[:3], raw_out, 7) == [0, 0, 0]  # no rectification: W x H results
    for k in range(3):
        want = np.zeros(w * h, np.float32)
        ctx.process_host(np.ascontiguousarray(capi.decode_gray8(blobs[idx[k]]).reshape(-1)), want, 7)
        assert bits_equal(raw_out[k], want), k


def test_image_pool_slabs():
    """ExposureImage's pool (csrc/host/image_pool.cpp): page-locked blocks come out of slabs of up to 64 images, lowest free
    address first, so the images made one after the other lie back to back; a block freed twice or a foreign pointer is ignored;
    slabs whose images are all back are released by the trim."""
    from mono_dataset_code_amd import capi

    L = capi.host_lib()
    L.mdch_image_alloc.restype = ctypes.c_void_p
    L.mdch_image_alloc.argtypes = [ctypes.c_ulong]
    L.mdch_image_free.argtypes = [ctypes.c_void_p]
    L.mdch_image_pool_trim()
    base_idle = L.mdch_image_pool_idle_bytes()
    n = 640 * 480 + 4096  # (a size no other test's images have: a slab shared with a reader that is still open would break the adjacency below)
    a = [L.mdch_image_alloc(n) for _ in range(128)]
    assert len(set(a)) == 128
    assert all(a[i + 1] - a[i] == n * 4 for i in range(63)), "one slab: 64 images back to back"
    assert all(a[i + 1] - a[i] == n * 4 for i in range(64, 127)), "the next slab"
    np.ctypeslib.as_array(ctypes.cast(a[127], ctypes.POINTER(ctypes.c_float)), (n,))[:] = 1.0  # writable to its end
    L.mdch_image_free(a[10])
    L.mdch_image_free(a[3])
    L.mdch_image_free(a[3])  # ignored
    L.mdch_image_free(a[3] + 64)  # not a block: ignored
    assert L.mdch_image_alloc(n) == a[3] and L.mdch_image_alloc(n) == a[10]  # (the only free blocks; lowest address first)
    assert L.mdch_image_pool_idle_bytes() == base_idle  # nothing releasable while images are live
    for p in a:
        L.mdch_image_free(p)
    assert L.mdch_image_pool_idle_bytes() == base_idle + 128 * n * 4
    L.mdch_image_pool_trim()
    assert L.mdch_image_pool_idle_bytes() == 0
    odd = [L.mdch_image_alloc(1001) for _ in range(3)]  # not whole 64-byte lines: blocks keep their alignment, not back to back
    assert odd[1] - odd[0] == 1008 * 4 and odd[0] % 64 == 0
    for p in odd:
        L.mdch_image_free(p)
    L.mdch_image_pool_trim()
    assert L.mdch_image_pool_idle_bytes() == 0


def islow_idct_reference(record, w, h, pitch, rows):
    """jidctint.c's `islow` inverse DCT on a coefficient record, restated with 64-bit numpy integers (the host decoder's
    arithmetic): quantisation table at the record's head, blocks of 64 int16 behind it -> h x w uint8."""
    q = record[:128].view(np.uint16).astype(np.int64).reshape(8, 8)
    blocks = record[128:128 + rows * pitch * 128].view(np.int16).reshape(rows, pitch, 8, 8).astype(np.int64) * q

    def one_d(c, shift):  # c[..., 8] along the last axis -> outputs along the last axis
        F = dict(a=2446, b=3196, c=4433, d=6270, e=7373, f=9633, g=12299, h=15137, i=16069, j=16819, k=20995, l=25172)
        z2, z3 = c[..., 2], c[..., 6]
        z1 = (z2 + z3) * F["c"]
        tmp2, tmp3 = z1 + z3 * -F["h"], z1 + z2 * F["d"]
        tmp0, tmp1 = (c[..., 0] + c[..., 4]) << 13, (c[..., 0] - c[..., 4]) << 13
        t10, t13, t11, t12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
        tmp0, tmp1, tmp2, tmp3 = c[..., 7], c[..., 5], c[..., 3], c[..., 1]
        z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
        z5 = (z3 + z4) * F["f"]
        tmp0, tmp1, tmp2, tmp3 = tmp0 * F["a"], tmp1 * F["j"], tmp2 * F["l"], tmp3 * F["g"]
        z1, z2, z3, z4 = z1 * -F["e"], z2 * -F["k"], z3 * -F["i"] + z5, z4 * -F["b"] + z5
        tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
        outs = [t10 + tmp3, t11 + tmp2, t12 + tmp1, t13 + tmp0, t13 - tmp0, t12 - tmp1, t11 - tmp2, t10 - tmp3]
        r = np.stack([(o + (1 << (shift - 1))) >> shift for o in outs], -1)
        return r.astype(np.int32).astype(np.int64)  # (the workspace is `int`: a 64-bit value is cut to 32 bits there, as on the host)

    ws = one_d(blocks.swapaxes(-1, -2), 11).swapaxes(-1, -2)  # pass 1 on columns
    px = np.clip(one_d(ws, 18) + 128, 0, 255).astype(np.uint8)  # pass 2 on rows
    img = px.transpose(0, 2, 1, 3).reshape(rows * 8, pitch * 8)
    return img[:h, :w]


@pytest.mark.parametrize("size", [(64, 96), (37, 70)])
def test_gpu_idct_on_coefficients_of_any_magnitude(size):
    """The device inverse DCT computes in 32 bits where no intermediate can leave them (inputs up to 35079: every real image) and
    in the host decoder's 64 bits otherwise.  A numpy restatement of the islow transform with 64-bit integers is first pinned to
    the host decoder on a real file, then both paths are driven with crafted records: coefficients and quantisation tables up to
    the full 16-bit range, magnitudes around the switch-over, single large terms among small ones."""
    import torch

    from mono_dataset_code_amd import capi

    h, w = size
    rec_bytes, pitch, rows = capi.jpeg_record_bytes(w, h)
    rng = np.random.default_rng(h * 131 + w)
    b = io.BytesIO()
    Image.fromarray(textured(h, w, 3)).save(b, "JPEG", quality=93)
    real = np.zeros(rec_bytes, np.uint8)
    capi.decode_jpeg_record(b.getvalue(), real, pitch)
    assert np.array_equal(islow_idct_reference(real, w, h, pitch, rows), capi.decode_gray8(b.getvalue()))
    recs = [real]
    nb = rows * pitch * 64
    for kind in range(8):
        r = np.zeros(rec_bytes, np.uint8)
        q = r[:128].view(np.uint16)
        c = r[128:128 + 2 * nb].view(np.int16)
        if kind == 0:    # everything anywhere
            q[:] = rng.integers(1, 65536, 64)
            c[:] = rng.integers(-32768, 32768, nb)
        elif kind == 1:  # around the switch-over: |coefficient * q| near 35079
            q[:] = 1
            c[:] = rng.integers(-32768, 32768, nb)
            c[rng.random(nb) < 0.5] //= 2
        elif kind == 2:  # small blocks with one huge term
            q[:] = rng.integers(1, 256, 64)
            c[:] = rng.integers(-20, 21, nb)
            idx = rng.integers(0, nb, nb // 40)
            c[idx] = rng.integers(-32768, 32768, len(idx))
        elif kind == 3:  # the largest magnitudes, all signs alike
            q[:] = 65535
            c[:] = 32767
        elif kind == 4:
            q[:] = 65535
            c[:] = -32768
        elif kind == 5:  # workspace values beyond the bound from coefficients below it
            q[:] = 8
            c[:] = rng.integers(-4300, 4301, nb)
        elif kind == 6:  # sparse blocks, as decoders mostly see them
            q[:] = rng.integers(1, 100, 64)
            c[:] = 0
            idx = rng.integers(0, nb, nb // 16)
            c[idx] = rng.integers(-300, 301, len(idx))
        else:            # alternating signs at full scale
            q[:] = 255
            c[:] = np.where(np.arange(nb) % 2 == 0, 2047, -2047)
        recs.append(r)
    n = len(recs)
    d_rec = torch.from_numpy(np.stack(recs)).cuda()
    d_frames = torch.full((n, h * w), 77, dtype=torch.uint8, device="cuda")
    ctx = capi.Context(0)
    ctx.jpeg_idct_batch(d_rec.data_ptr(), rec_bytes, d_frames.data_ptr(), w, h, pitch, rows, n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_frames.cpu().numpy().reshape(n, h, w)
    for i in range(n):
        want = islow_idct_reference(recs[i], w, h, pitch, rows)
        assert np.array_equal(got[i], want), (i, int((got[i] != want).sum()), np.argwhere(got[i] != want)[:3].tolist())


@pytest.mark.parametrize("fmt,stage", [("jpg", 2), ("jpg", 1), ("jpg", 0), ("png", 2)])
def test_get_images_device_equals_get_images(tmp_path, oracle, fmt, stage):
    """DatasetReader::getImagesDevice (row f1/f2/f4 with a device-resident end): files -> decode pool -> upload -> (Huffman ->) (inverse
    DCT ->) fused pass, results LEFT IN HBM in the caller's arrays.  After a copy back: the same bytes as getImages, for every JPEG
    stage and for PNG, sub-ranges included; with levels = 4 and gradient arrays: == mdc_process_pyramid_gradients_batch_device on the
    decoded frames == the oracle's chain (box levels and gradients: own definition, parity unpinned)."""
    import torch

    from mono_dataset_code_amd import capi

    h, w = 256, 320
    n = 70  # more than one 64-frame chunk of the stream pipeline
    frames = frames_for(n, h, w)
    names, blobs = make_sequence(str(tmp_path), frames, True, fmt)
    r = capi.DatasetReader(str(tmp_path))
    r.set_gpu_jpeg(stage)
    assert r.device() == 0
    ow, oh = r.out_w, r.out_h
    for first, count, fl in ((0, n, (1, 1, 1, 1)), (5, 40, (0, 1, 1, 0)), (3, 66, (1, 0, 0, 1))):
        want, ok, got = r.get_images(first, count, *fl)
        assert got == count and ok.all()
        npo = ow * oh if fl[0] else w * h
        d_base = torch.full((count, npo), -3.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()  # (the call runs on the reader's own streams)
        outs = capi.DeviceOutputs.make(d_base.data_ptr())
        valid, got_d = r.get_images_device(first, count, *fl, outs)
        assert got_d == count and valid.all(), r.last_error()
        assert bits_equal(d_base.cpu().numpy(), want), (fmt, stage, first, count, fl)
    # base + levels 1..3 + gradient images of every level, rectified
    count, fl = n, (1, 1, 1, 1)
    dims = [(ow >> l, oh >> l) for l in range(4)]
    d_base = torch.zeros((count, ow * oh), dtype=torch.float32, device="cuda")
    d_lv = [torch.zeros((count, a * b), dtype=torch.float32, device="cuda") for a, b in dims[1:]]
    d_dI = [torch.zeros((count, a * b * 3), dtype=torch.float32, device="cuda") for a, b in dims]
    d_ab = [torch.zeros((count, a * b), dtype=torch.float32, device="cuda") for a, b in dims]
    torch.cuda.synchronize()
    outs = capi.DeviceOutputs.make(d_base.data_ptr(), 4, [t.data_ptr() for t in d_lv], [t.data_ptr() for t in d_dI], [t.data_ptr() for t in d_ab])
    valid, got_d = r.get_images_device(0, count, *fl, outs)
    assert got_d == count and valid.all(), r.last_error()
    want, ok, got = r.get_images(0, count, *fl)
    assert bits_equal(d_base.cpu().numpy(), want)
    for f in (0, 63, 64, n - 1):
        src, cw, ch = want[f], ow, oh
        for l in range(4):
            if l:
                src = oracle.pyramid_level(src, cw, ch)
                cw, ch = cw // 2, ch // 2
                assert bits_equal(d_lv[l - 1][f].cpu().numpy(), src), (f, l)
            w_dI, w_abs = oracle.gradients(src, cw, ch)
            assert bits_equal(d_dI[l][f].cpu().numpy(), w_dI.reshape(-1)), (f, l, "dI")
            assert bits_equal(d_ab[l][f].cpu().numpy(), w_abs), (f, l, "abs")
    r.close()


def test_process_frames_host_to_device_with_frame_index(tmp_path, oracle):
    """mdc_process_frames_host_to_device through the C ABI: raw host frames -> device array positions given by frame_index (scattered,
    runs of consecutive positions batched); mdc_device_alloc / mdc_copy_to_host for callers without a HIP toolchain."""
    from mono_dataset_code_amd import capi, synth

    d = synth.write_sequence_calibration(str(tmp_path), synth.camera_lines(320, 256, 160, 120))
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    photo = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), 320, 256)
    ctx = capi.Context(0)
    ctx.bind(fov, photo)
    n, npi, npo = 9, 320 * 256, 160 * 120
    raws = [f for f in synth.noise_frames(0, n, npi)]
    flags = capi.GAMMA | capi.VIGNETTE | capi.KILL_OVEREXPOSED | capi.RECTIFY
    L = capi.hip_lib()
    d_ptr = ctypes.c_void_p()
    positions = 12
    assert L.mdc_device_alloc(ctx.handle, positions * npo * 4, ctypes.byref(d_ptr)) == 0
    index = [3, 4, 5, 0, 11, 10, 7, 8, 1]
    ctx.process_frames_host_to_device(raws, flags, capi.DeviceOutputs.make(d_ptr.value), index)
    got = np.zeros((positions, npo), np.float32)
    assert L.mdc_copy_to_host(ctx.handle, got.ctypes.data_as(ctypes.c_void_p), d_ptr, got.nbytes) == 0
    rx, ry = fov.remap()
    for i, pos in enumerate(index):
        want = oracle.get_image(raws[i], 320, 256, 160, 120, photo.ginv(), photo.vignette()[1], True, True, rx, ry, 1, 1, 1, 1)
        assert bits_equal(got[pos], want), (i, pos)
    L.mdc_device_free(ctx.handle, d_ptr)
    with pytest.raises(capi.MdcError):  # no outputs / a negative position: refused
        ctx.process_frames_host_to_device(raws, flags, capi.DeviceOutputs.make(0))
    with pytest.raises(capi.MdcError):
        ctx.process_frames_host_to_device(raws[:1], flags, capi.DeviceOutputs.make(1), [-1])
    ctx.close()


@pytest.mark.parametrize("fmt", ["png", "jpg"])
def test_get_images_device_longer_than_the_ring(tmp_path, fmt):
    """getImagesDevice over more frames than a lane's page-locked ring holds, with both lanes of the output device at work (the second
    context is made at the first call): ring slots are re-used while the pool decodes ahead; every position equals getImages'."""
    import torch

    from mono_dataset_code_amd import capi

    h, w = 64, 80
    n = 700
    frames = [textured(h, w, s % 37) for s in range(n)]
    make_sequence(str(tmp_path), frames, True, fmt)
    r = capi.DatasetReader(str(tmp_path))
    want, ok, got = r.get_images(0, n, 1, 1, 1, 0)
    assert got == n and ok.all()
    for threads in (0, 3):
        r.set_threads(threads)
        d_base = torch.full((n, r.out_w * r.out_h), -3.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        valid, got_d = r.get_images_device(0, n, 1, 1, 1, 0, capi.DeviceOutputs.make(d_base.data_ptr()))
        assert got_d == n and valid.all(), r.last_error()
        assert bits_equal(d_base.cpu().numpy(), want), (fmt, threads)
    st = r.device_stats()  # the twin context getImagesDevice made on device 0 is not a device of its own: one entry, its frames folded in
    assert len(st) == 1 and st[0][0] == 0 and st[0][1] >= 3 * n, st
    want2, ok2, got2 = r.get_images(10, 50, 1, 1, 1, 0)
    assert got2 == 50 and bits_equal(want2, want[10:60])
    r.close()
