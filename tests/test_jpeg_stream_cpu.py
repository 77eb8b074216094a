"""mdch_jpeg_stream (the host half of the device Huffman path: markers parsed, the decode tables built, byte stuffing and
restart markers removed, restart intervals located; one component or YCbCr interleaved) checked WITHOUT a GPU: a plain sequential decoder written here from the table format include/mdc_hip.h documents
(mdc_jpeg_huff: t1 by the next 11 bits, 32-entry subtables for the codes of 12..16 bits) turns the stream back into a
coefficient record, which must equal the host decoder's record (mdch_decode_jpeg_record, itself pinned on libjpeg by
tests/test_reader_cpu.py) byte for byte.  The kernel (csrc/mdc_jpeg.hip) reads the same tables with the same rules; its own
parity tests are the gpu ones in tests/test_reader.py."""
import io

import numpy as np
import pytest
from PIL import Image

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


def textured(h, w, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = 120 + 70 * np.sin(xx * 0.11 + seed) * np.cos(yy * 0.07) + rng.normal(0, 12, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def parse_header(stream):
    """-> dict of the header fields, the decode tables ([0] DC luma, [1] AC luma, [2] DC chroma, [3] AC chroma) and the restart
    intervals' start bytes (include/mdc_hip.h: mdc_jpeg_stream_header and what follows it)."""
    from mono_dataset_code_amd import capi

    u32 = stream[:32].view(np.uint32)
    magic, w, h, ecs, ri, n_iv, comp_info, ecs_off = (int(x) for x in u32[:8])
    assert magic == 0x33534A4D
    ncomp, hy, vy = comp_info & 255, (comp_info >> 8) & 15, (comp_info >> 12) & 15
    assert ncomp in (1, 3) and (ncomp == 3 or (hy, vy) == (1, 1)) and ecs_off % 16 == 0 and n_iv >= 1 and (ri or n_iv == 1)
    quant = stream[32:160].view(np.uint16).copy()
    ntab = 4 if ncomp == 3 else 2
    assert capi.JPEG_STREAM_HEADER_BYTES == 160 + 2 * 12288
    tabs = stream[160:160 + ntab * 12288].view(np.uint32).reshape(ntab, 2048 + 32 * 32)
    t1 = [tabs[k, :2048] for k in range(ntab)]
    t2 = [tabs[k, 2048:].reshape(32, 32) for k in range(ntab)]
    at = 160 + ntab * 12288
    starts = [int(x) for x in stream[at:at + 4 * n_iv].view(np.uint32)] if ri else [0]
    assert ecs_off >= at + (4 * n_iv if ri else 0) and ecs_off - (at + (4 * n_iv if ri else 0)) < 16
    return dict(w=w, h=h, ecs=ecs, ri=ri, n_iv=n_iv, ncomp=ncomp, hy=hy, vy=vy, ecs_off=ecs_off, quant=quant, t1=t1, t2=t2, starts=starts)


def sequential_decode(stream, pitch, rows):
    """-> luma record (uint8) decoded one symbol after the other with the stream's own tables: MCU by MCU (hY x vY luma blocks,
    then Cb, then Cr for three components), restart interval by restart interval (each from its recorded start byte, predictors 0)."""
    H = parse_header(stream)
    w, h, ecs, t1, t2 = H["w"], H["h"], H["ecs"], H["t1"], H["t2"]
    off = H["ecs_off"]
    assert not stream[off + ecs: off + ecs + 16].any(), "16 zero bytes follow the entropy-coded segment"
    data = bytes(stream[off: off + ecs + 16])
    total = len(data) * 8
    val = int.from_bytes(data, "big")

    def peek(p, n):
        return (val >> (total - p - n)) & ((1 << n) - 1) if n else 0

    hy, vy = H["hy"], H["vy"]
    mx, my = (w + 8 * hy - 1) // (8 * hy), (h + 8 * vy - 1) // (8 * vy)
    nb = hy * vy + (2 if H["ncomp"] == 3 else 0)
    rec = np.zeros(128 + rows * pitch * 128, np.uint8)
    rec[:128] = H["quant"].view(np.uint8)
    coef = rec[128:].view(np.int16).reshape(rows, pitch, 64)
    scratch = np.zeros(64, np.int16)
    mcus = mx * my
    ri = H["ri"] or mcus
    assert H["n_iv"] == (mcus + ri - 1) // ri
    p = 0
    for iv in range(H["n_iv"]):
        p = H["starts"][iv] * 8
        end = (H["starts"][iv + 1] if iv + 1 < H["n_iv"] else ecs) * 8
        dc = [0, 0, 0]
        for m in range(iv * ri, min(mcus, (iv + 1) * ri)):
            for u in range(nb):
                luma = u < hy * vy
                c = 0 if luma else 1 + (u - hy * vy)
                tab = 0 if luma else 2
                out = coef[(m // mx) * vy + u // hy, (m % mx) * hy + u % hy] if luma else scratch
                z = 0
                while z < 64:
                    ac = 1 if z else 0
                    e = int(t1[tab + ac][peek(p, 11)])
                    if e & 31 == 31:
                        e = int(t2[tab + ac][(e >> 16) & 31][peek(p + 11, 5)])
                    ln, run, size = e & 31, (e >> 5) & 15, (e >> 9) & 15
                    assert 1 <= ln <= 16, (m, u, z, hex(e))
                    if e & (1 << 13):
                        v = (e >> 16) - (1 << 16) if e >> 31 else e >> 16
                        p += ln + size
                    else:
                        p += ln
                        bits = peek(p, size)
                        p += size
                        v = bits if size == 0 or bits >> (size - 1) else bits - (1 << size) + 1
                    if not ac:
                        dc[c] += v
                        out[0] = dc[c]
                        z = 1
                    elif size == 0:
                        z = z + 16 if run == 15 else 64
                    else:
                        z += run
                        assert z < 64
                        out[ZIGZAG[z]] = v
                        z += 1
        assert p <= end and end - p < 8 + 8, "an interval ends inside its last byte (padding bits only)"
    return rec


@pytest.mark.parametrize("size", [(64, 64), (100, 130), (17, 23), (8, 8)])
def test_stream_tables_decode_to_the_host_decoders_record(size):
    from mono_dataset_code_amd import capi

    h, w = size
    rng = np.random.default_rng(h * 7 + w)
    imgs = [textured(h, w, 4), rng.integers(0, 256, (h, w), dtype=np.uint8), np.full((h, w), 131, np.uint8)]
    rec_bytes, pitch, rows = capi.jpeg_record_bytes(w, h)
    bw, bh = (w + 7) // 8, (h + 7) // 8
    for k, img in enumerate(imgs):
        rgb = np.stack([img, np.roll(img, 3, 1), 255 - img], -1)
        for kw in ({"quality": 5}, {"quality": 50}, {"quality": 92}, {"quality": 100}, {"quality": 75, "optimize": True},
                   # restart intervals (DRI + RSTn markers), one component and three
                   {"quality": 80, "restart_marker_blocks": 7}, {"quality": 90, "restart_marker_rows": 1}, {"quality": 60, "restart_marker_blocks": 1},
                   # YCbCr, the three components interleaved: 4:4:4, 4:2:2, 4:2:0 -- only the luma plane is decoded
                   {"quality": 88, "subsampling": 0, "rgb": 1}, {"quality": 70, "subsampling": 1, "rgb": 1}, {"quality": 93, "subsampling": 2, "rgb": 1},
                   {"quality": 85, "subsampling": 2, "rgb": 1, "restart_marker_rows": 1}, {"quality": 40, "subsampling": 1, "rgb": 1, "restart_marker_blocks": 3, "optimize": True}):
            kw = dict(kw)
            src = rgb if kw.pop("rgb", 0) else img
            b = io.BytesIO()
            Image.fromarray(src).save(b, "JPEG", **kw)
            data = b.getvalue()
            want = np.zeros(rec_bytes, np.uint8)
            assert capi.decode_jpeg_record(data, want, pitch)[:3] == (w, h, pitch)
            stream = np.full((2 * capi.JPEG_STREAM_HEADER_BYTES + 4 * bw * bh + len(data) + 64 + 15) & ~15, 0xA5, np.uint8)
            used, sw, sh = capi.jpeg_stream(data, stream)
            assert (sw, sh) == (w, h) and used <= stream.size
            assert (stream[used:] == 0xA5).all(), "nothing written beyond the bytes reported"
            got = sequential_decode(stream, pitch, rows)
            assert np.array_equal(got[:128], want[:128]), (k, kw, "quantisation table")
            g = got[128:].view(np.int16).reshape(rows, pitch, 64)[:bh, :bw]
            e = want[128:rec_bytes].view(np.int16).reshape(rows, pitch, 64)[:bh, :bw]
            assert np.array_equal(g, e), (k, kw)


def test_stream_table_entries_are_well_formed():
    """Every t1 entry is either empty (length 0), a code of 1..11 bits whose fields fit their ranges, or a pointer to one of
    the subtables in use; value entries (bit 13) only where code + magnitude bits fit the 11-bit window and hold what
    decoding the magnitude bits would give; subtable entries have lengths 12..16 or 0."""
    from mono_dataset_code_amd import capi

    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (96, 96), dtype=np.uint8)
    for kw in ({"quality": 30}, {"quality": 97}, {"quality": 60, "optimize": True}):
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", **kw)
        stream = np.zeros((capi.JPEG_STREAM_HEADER_BYTES + len(b.getvalue()) + 64 + 15) & ~15, np.uint8)
        capi.jpeg_stream(b.getvalue(), stream)
        hdr = parse_header(stream)
        t1, t2 = hdr["t1"], hdr["t2"]
        for ac in (0, 1):
            subs = set()
            for idx in range(2048):
                e = int(t1[ac][idx])
                ln, run, size = e & 31, (e >> 5) & 15, (e >> 9) & 15
                if ln == 31:
                    assert (e >> 16) < 32
                    subs.add(e >> 16)
                    continue
                assert ln <= 11
                if ln == 0:
                    continue
                if not ac:
                    assert run == 0 and size <= 11
                if e & (1 << 13):
                    assert ln + size <= 11
                    bits = (idx >> (11 - ln - size)) & ((1 << size) - 1)
                    v = bits if size == 0 or bits >> (size - 1) else bits - (1 << size) + 1
                    assert ((e >> 16) - (1 << 16) if e >> 31 else e >> 16) == v
            for s in range(32):
                for e in t2[ac][s]:
                    ln = int(e) & 31
                    assert ln == 0 or (12 <= ln <= 16 and s in subs), (ac, s, hex(int(e)))
                    assert not int(e) & (1 << 13)


def test_stream_refusals_and_small_buffers():
    from mono_dataset_code_amd import capi

    img = textured(48, 40, 2)
    rgb = np.stack([img, np.roll(img, 3, 1), 255 - img], -1)
    big = np.zeros(1 << 17, np.uint8)
    # refused (the caller takes the host decoders): progressive files; restart markers that do not count up
    for im, kw in ((rgb, {"quality": 88, "progressive": True}), (img, {"quality": 85, "progressive": True})):
        b = io.BytesIO()
        Image.fromarray(im).save(b, "JPEG", **kw)
        with pytest.raises(ValueError):
            capi.jpeg_stream(b.getvalue(), big)
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=80, restart_marker_blocks=2)
    bad = bytearray(b.getvalue())
    at = bad.index(b"\xff\xd1")
    bad[at + 1] = 0xD5
    with pytest.raises(ValueError):
        capi.jpeg_stream(bytes(bad), big)
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=90)
    data = b.getvalue()
    used = capi.jpeg_stream(data, big)[0]
    small = np.full(used, 0xA5, np.uint8)
    with pytest.raises(ValueError):
        capi.jpeg_stream(data, small[: used - 16])  # a buffer too small is refused, not overrun
    assert (small[used - 16:] == 0xA5).all()
    # a scan cut short is passed on as it is (fewer bits; the device reports status 1 = too few blocks and the reader hands the
    # file to the host decoder, tests/test_reader.py::test_reader_gpu_jpeg_stages_agree_on_damaged_and_mixed_files)
    cut = np.zeros(1 << 17, np.uint8)
    capi.jpeg_stream(data[: len(data) - 200], cut)
    assert parse_header(cut)["ecs"] < parse_header(big)["ecs"]
    with pytest.raises(ValueError):
        capi.jpeg_stream(b"\xff\xd8\xff\xd9", big)
    with pytest.raises(ValueError):
        capi.jpeg_stream(b"", big)


def test_stream_tables_are_cached_per_table_definition():
    """mdch_jpeg_stream keeps the device tables of the last table definitions it saw (per thread): files with the encoder's default
    tables and files with optimised tables in alternation must each get their own tables -- equal to what a first call builds --
    and decode to the host decoder's record."""
    from mono_dataset_code_amd import capi

    h, w = 64, 80
    imgs = [textured(h, w, s) for s in range(3)]
    files = []
    for k, img in enumerate(imgs):
        for kw in ({"quality": 90}, {"quality": 90, "optimize": True}, {"quality": 35, "optimize": True}):
            b = io.BytesIO()
            Image.fromarray(img).save(b, "JPEG", **kw)
            files.append(b.getvalue())
    rec_bytes, pitch, rows = capi.jpeg_record_bytes(w, h)
    bw, bh = (w + 7) // 8, (h + 7) // 8
    first = {}
    for order in (range(len(files)), reversed(range(len(files))), [0, 0, 1, 0, 4, 4, 8, 1]):
        for i in order:
            stream = np.zeros((capi.JPEG_STREAM_HEADER_BYTES + len(files[i]) + 64 + 15) & ~15, np.uint8)
            used = capi.jpeg_stream(files[i], stream)[0]
            if i in first:
                assert np.array_equal(stream[:used], first[i]), i
                continue
            first[i] = stream[:used].copy()
            want = np.zeros(rec_bytes, np.uint8)
            capi.decode_jpeg_record(files[i], want, pitch)
            got = sequential_decode(stream, pitch, rows)
            g = got[128:].view(np.int16).reshape(rows, pitch, 64)[:bh, :bw]
            e = want[128:rec_bytes].view(np.int16).reshape(rows, pitch, 64)[:bh, :bw]
            assert np.array_equal(got[:128], want[:128]) and np.array_equal(g, e), i
