"""ASan + UBSan over the host layer: tests/native/host_sanitize.cpp is compiled together with
the host sources under -fsanitize=address,undefined and run on valid, malformed, truncated and bit-flipped
calibration files, images and zip archives, plus the reader's decode pool.  No GPU involved."""
import io
import os
import sys
import subprocess
import zipfile

import numpy as np
import pytest
from PIL import Image

from test_reader_cpu import make_sequence, textured


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def make_fixtures(tmp_path):
    """valid, malformed, truncated and bit-flipped inputs for the harnesses; -> root folder"""
    from mono_dataset_code_amd import synth

    root = tmp_path / "fix"
    for d in ("images_any", "calib", "vignettes", "zips", "sequences"):
        (root / d).mkdir(parents=True)
    img = textured(70, 90, 1)
    Image.fromarray(img).save(root / "images_any" / "a.png")
    Image.fromarray(img).save(root / "images_any" / "b.jpg", quality=85)
    Image.fromarray(img).save(root / "images_any" / "c_opt.jpg", quality=40, optimize=True)
    Image.fromarray(img).save(root / "images_any" / "d_rst.jpg", quality=90, restart_marker_blocks=5)
    Image.fromarray(np.stack([img, img[::-1], 255 - img], -1)).save(root / "images_any" / "e_color420.jpg", quality=80, subsampling=2)
    Image.fromarray(img).save(root / "images_any" / "f_prog.jpg", progressive=True)
    Image.fromarray(img.astype(np.uint16) << 8).save(root / "images_any" / "g16.png")
    open(root / "images_any" / "h.pgm", "wb").write(b"P5\n# c\n90 70\n255\n" + img.tobytes())
    open(root / "images_any" / "i.pgm", "wb").write(b"P5 99999 99999 255\n" + bytes(100))
    open(root / "images_any" / "j.bin", "wb").write(os.urandom(3000))
    # the extended decoders (image_codecs_ext.cpp): every PNG flavour, Adam7, progressive JPEG -- sound, truncated and bit-flipped
    from test_reader_cpu import interlaced_png, png_bytes, raw_png

    rgb = np.stack([img, img[::-1], 255 - img], -1)
    prog = open(root / "images_any" / "f_prog.jpg", "rb").read()
    extra = {"k_rgb.png": png_bytes(rgb), "l_pal.png": png_bytes(Image.fromarray(rgb).quantize(colors=13), bits=4),
             "m_lace.png": interlaced_png(img), "n_lace16rgb.png": interlaced_png((rgb.astype(np.uint16) << 8)),
             "o_rgb16.png": raw_png(rgb.astype(np.uint16) * 257), "p_prog_color.jpg": None}
    b = __import__("io").BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", quality=80, progressive=True, subsampling=2, restart_marker_blocks=4)
    extra["p_prog_color.jpg"] = b.getvalue()
    for k, v in extra.items():
        open(root / "images_any" / k, "wb").write(v)
        open(root / "images_any" / ("t_" + k), "wb").write(v[: len(v) * 3 // 5])
        for j, at in enumerate((len(v) // 3, len(v) // 2, len(v) - 40)):
            f = bytearray(v)
            f[at] ^= 0x5A
            open(root / "images_any" / ("f%d_" % j + k), "wb").write(bytes(f))
    open(root / "images_any" / "t_prog.jpg", "wb").write(prog[: len(prog) // 2])
    # calibration files: valid variants + malformed ones (as tests/test_tables_vs_ref.py uses)
    cams = {"camera_ok.txt": "0.349153 0.436593 0.493140 0.499021 0.933271\n48 32\ncrop\n30 20\n",
            "camera_full.txt": "0.349153 0.436593 0.493140 0.499021 0.933271\n48 32\nfull\n30 20\n",
            "camera_none.txt": "0.349153 0.436593 0.493140 0.499021 0.933271\n48 32\nnone\n30 20\n",
            "camera_explicit.txt": "0.5 0.6 0.5 0.5 0\n48 32\n0.4 0.53 0.5 0.5 0\n31 21\n",
            "camera_short.txt": "0.349153 0.436593\n", "camera_empty.txt": "", "camera_words.txt": "a b c d e\nx y\ncrop\n1 1\n",
            "camera_huge.txt": "0.3 0.4 0.5 0.5 0.9\n48 32\ncrop\n0 0\n", "camera_neg.txt": "0.3 0.4 0.5 0.5 0.9\n-5 32\ncrop\n30 20\n"}
    for k, v in cams.items():
        open(root / "calib" / k, "w").write(v)
    good = " ".join("%.6f" % (255.0 * (i / 255.0) ** 2.2 + 0.01 * i) for i in range(256))
    for k, v in {"pcalib_ok.txt": good + "\n", "pcalib_255.txt": " ".join(good.split()[:255]) + "\n", "pcalib_flat.txt": " ".join(["1.0"] * 256) + "\n",
                 "pcalib_empty.txt": "", "pcalib_text.txt": "hello world\n"}.items():
        open(root / "calib" / k, "w").write(v)
    v16 = np.asarray(synth.vignette_image(48, 32, 16)).reshape(32, 48).astype(np.uint16)
    synth.write_png_gray(str(root / "vignettes" / "v16.png"), v16)
    synth.write_png_gray(str(root / "vignettes" / "v8.png"), (v16 >> 8).astype(np.uint8))
    synth.write_png_gray(str(root / "vignettes" / "wrong_size.png"), np.asarray(synth.vignette_image(40, 32, 16)).reshape(32, 40).astype(np.uint16))
    open(root / "vignettes" / "garbage.png", "wb").write(b"\x89PNG\r\n\x1a\n" + os.urandom(200))
    # zips
    blobs = [open(root / "images_any" / n, "rb").read() for n in ("a.png", "b.jpg")]
    for name, method in (("stored.zip", zipfile.ZIP_STORED), ("deflated.zip", zipfile.ZIP_DEFLATED)):
        with zipfile.ZipFile(root / "zips" / name, "w", method) as z:
            for i, b in enumerate(blobs):
                z.writestr("%02d.bin" % i, b)
    whole = open(root / "zips" / "deflated.zip", "rb").read()
    open(root / "zips" / "truncated.zip", "wb").write(whole[: len(whole) * 2 // 3])
    open(root / "zips" / "tail_only.zip", "wb").write(whole[-60:])
    flipped = bytearray(whole)
    flipped[len(whole) // 3] ^= 0xFF
    open(root / "zips" / "flipped.zip", "wb").write(bytes(flipped))
    open(root / "zips" / "garbage.zip", "wb").write(os.urandom(500))
    # hostile 64-bit / size fields: range checks must not wrap, sizes must not become allocations
    import struct
    stored = open(root / "zips" / "stored.zip", "rb").read()
    eocd = stored.rfind(b"PK\x05\x06")
    cd_off = struct.unpack_from("<I", stored, eocd + 16)[0]
    # (1) ZIP64 EOCD whose cd_off + cd_size wraps around 2^64
    z64 = struct.pack("<IQHHIIQQQQ", 0x06064B50, 44, 45, 45, 0, 0, 2, 2, 20, 2**64 - 10)
    loc = struct.pack("<IIQI", 0x07064B50, 0, eocd, 1)
    open(root / "zips" / "zip64_wrap.zip", "wb").write(stored[:eocd] + z64 + loc + stored[eocd:])
    # (2) a central-directory entry that declares 0xfffffff0 uncompressed bytes
    big = bytearray(stored)
    struct.pack_into("<I", big, cd_off + 24, 0xFFFFFFF0)
    open(root / "zips" / "huge_usize.zip", "wb").write(bytes(big))
    # (3) a ZIP64 extra field in the LAST entry whose length runs past the extra area (and the file)
    ent = bytearray(stored[cd_off:eocd])
    last = ent.rfind(b"PK\x01\x02")
    nl = struct.unpack_from("<H", ent, last + 28)[0]
    struct.pack_into("<III", ent, last + 20, 0xFFFFFFFF, 0xFFFFFFFF, 0)  # csize, usize = "see ZIP64 extra"
    struct.pack_into("<I", ent, last + 42, 0xFFFFFFFF)
    struct.pack_into("<H", ent, last + 30, 4)  # extra length: just the 4-byte field header
    extra = struct.pack("<HH", 1, 0xFFFF)
    ent = ent[: last + 46 + nl] + extra
    tail = bytearray(stored[eocd:])
    struct.pack_into("<I", tail, 12, len(ent))
    open(root / "zips" / "zip64_extra_overrun.zip", "wb").write(stored[:cd_off] + bytes(ent) + bytes(tail))
    # sequences (folder + zip), one with a corrupt frame
    frames = [textured(32, 48, s) for s in range(6)]
    for name, zipped, fmt in (("seq_png", False, "png"), ("seq_zip_jpg", True, "jpg")):
        (root / "sequences" / name).mkdir()
        make_sequence(str(root / "sequences" / name), frames, zipped, fmt)
    open(root / "sequences" / "seq_png" / "images" / "00003.png", "wb").write(b"\x89PNG\r\n\x1a\n" + os.urandom(64))
    # a sequence archive with one implausible size field: the decode POOL's threads meet it (must report a bad frame, not terminate)
    (root / "sequences" / "seq_zip_badsize").mkdir()
    make_sequence(str(root / "sequences" / "seq_zip_badsize"), frames, True, "png")
    zp = root / "sequences" / "seq_zip_badsize" / "images.zip"
    zb = bytearray(open(zp, "rb").read())
    e0 = zb.rfind(b"PK\x05\x06")
    c0 = struct.unpack_from("<I", zb, e0 + 16)[0]
    struct.pack_into("<I", zb, c0 + 24, 0xFFFFFFF0)
    open(zp, "wb").write(bytes(zb))

    return root


def test_host_layer_under_asan_ubsan(tmp_path):
    from mono_dataset_code_amd import build, synth

    exe = str(tmp_path / "host_sanitize")
    # skip only where the toolchain has no sanitizer runtime; a compile error in OUR sources must fail the test
    probe = tmp_path / "probe.cpp"
    probe.write_text("int main() { return 0; }\n")
    if subprocess.run(["g++", "-fsanitize=address,undefined", str(probe), "-o", str(tmp_path / "probe")], stdout=subprocess.PIPE,
                      stderr=subprocess.STDOUT).returncode != 0:
        pytest.skip("no sanitizer runtime in this toolchain")
    build.build_host_sanitize(exe)
    root = make_fixtures(tmp_path)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(root)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    tail = r.stdout[-4000:]
    assert "AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout, tail
    assert r.returncode == 0 and "HOST_SANITIZE_OK" in r.stdout, tail


def test_host_threads_under_tsan(tmp_path):
    """ThreadSanitizer over the host layer's threads (tests/native/host_tsan.cpp): the reader's decode pool with the caller
    consuming, jumping, resizing the pool and dying with prefetches in flight; four readers on four threads; the ExposureImage
    pool from eight threads with a concurrent trim; the decoders and mdch_jpeg_stream on shared input."""
    from mono_dataset_code_amd import build

    probe = tmp_path / "probe.cpp"
    probe.write_text("#include <thread>\nint main() { std::thread t([] {}); t.join(); return 0; }\n")
    if subprocess.run(["g++", "-fsanitize=thread", str(probe), "-lpthread", "-o", str(tmp_path / "probe")], stdout=subprocess.PIPE,
                      stderr=subprocess.STDOUT).returncode != 0 or subprocess.run([str(tmp_path / "probe")]).returncode != 0:
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    exe = str(tmp_path / "host_tsan")
    build.build_host_tsan(exe)
    root = make_fixtures(tmp_path)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1")
    r = subprocess.run([exe, str(root)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    tail = r.stdout[-6000:]
    assert "ThreadSanitizer" not in r.stdout, tail
    assert r.returncode == 0 and "HOST_TSAN_OK" in r.stdout, tail
