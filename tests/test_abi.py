"""The C-ABI shared libraries load and export every symbol their headers declare
(no compute calls: this runs without a GPU), and without a GPU the product fails
loudly instead of falling back to a CPU implementation."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_libmdc_hip_exports_header():
    from mono_dataset_code_amd import capi

    names = declared("mdc_hip.h", "mdc_")
    assert sorted(capi.HIP_SYMBOLS) == names
    lib = ctypes.CDLL(capi.LIB_HIP_PATH)
    for n in names:
        assert hasattr(lib, n), n


def exported(lib):
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_exported_symbols_are_exactly_the_headers():
    """-fvisibility=hidden + MDC_API + a version script: the dynamic symbol table of every C-ABI library is the set of functions its
    header declares, nothing more (round 4 exported ~70 internal mdc:: C++ symbols beside the ABI)."""
    import subprocess

    from mono_dataset_code_amd import build

    assert exported(build.LIB_HIP) == declared("mdc_hip.h", "mdc_")
    assert exported(build.LIB_MULTI) == declared("mdc_multi.h", "mdc_multi_")
    assert exported(build.LIB_BENCH) == declared("mdc_bench.h", "mdcb_")
    # the host library: the C facade + the reference's classes and the two binding helpers (C++ linkage), nothing internal
    host = exported(build.LIB_HOST)
    assert sorted(n for n in host if not n.startswith("_Z")) == declared("mdc_host.h", "mdch_")
    dem = subprocess.run(["c++filt"], input="\n".join(n for n in host if n.startswith("_Z")), stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()
    for d in dem:
        assert re.match(r"(void )?(UndistorterFOV|PhotometricUndistorter|DatasetReader|ExposureImage)::|mdc_bind_objects\(|mdc_fov_model_of\(", d), d
    for txt in (open(os.path.join(ROOT, "include", h)).read() for h in ("mdc_hip.h", "mdc_host.h", "mdc_multi.h", "mdc_bench.h")):
        for line in txt.splitlines():  # every declaration carries the marker
            if re.match(r"(const char\*|int64_t|int|void\*?|long long|float\*?|unsigned long|size_t|double|mdch?_\w+\*)\s+mdc\w+\(", line):
                raise AssertionError("declaration without MDC_API: " + line)


def test_code_id_ignores_comments_only():
    """build.code_text: comments and spacing are not part of a build's identity; string literals and every token are."""
    from mono_dataset_code_amd import build
    a = 'int f() { return g("// no comment /* here */"); }  // why\n/* a block\n   of text */ int x = \'"\';\n'
    b = '  int f() { return g("// no comment /* here */"); }\n\n int x = \'"\';   // another remark\n'
    assert build.code_text(a) == build.code_text(b)
    assert build.code_text(a) != build.code_text(a.replace("return g", "return h"))
    assert build.code_text(a) != build.code_text(a.replace("no comment", "no  comment"))


def test_code_id_names_the_build():
    """mdc_code_id(): the hash of sources + flags the library was made from == what the build recipe computes for the tree."""
    from mono_dataset_code_amd import build, capi

    assert re.fullmatch(r"[0-9a-f]{16}", capi.code_id())
    assert capi.code_id() == build.code_id()
    import ctypes

    dbg = ctypes.CDLL(build.build_debug())
    dbg.mdc_code_id.restype = ctypes.c_char_p
    assert dbg.mdc_code_id().decode() == build.code_id(["MDC_DEBUG_BOUNDS=1"]) != capi.code_id()


def test_stale_traffic_entries_are_refused(tmp_path):
    """bench.py quotes profiles/hbm_traffic.json only for the kernel instantiation AND the build it was measured on."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import bench

    p = tmp_path / "hbm_traffic.json"
    p.write_text(json.dumps({"fused:k<1>": {"kernel": "k<1>", "bytes_per_frame": 100.0, "code_id": "aaaaaaaaaaaaaaaa", "source": "profiles/x.json"},
                             "old:k<2>": {"kernel": "k<2>", "bytes_per_frame": 7.0, "source": "profiles/old.json"}}))
    assert bench.traffic_from_profiles("k<1>", 10, "aaaaaaaaaaaaaaaa", str(p)) == (1000, "profiles/x.json")
    t, why = bench.traffic_from_profiles("k<1>", 10, "bbbbbbbbbbbbbbbb", str(p))
    assert t is None and "stale" in why and "aaaaaaaaaaaaaaaa" in why and "bbbbbbbbbbbbbbbb" in why
    t, why = bench.traffic_from_profiles("k<2>", 10, "bbbbbbbbbbbbbbbb", str(p))  # recorded before builds were identified: refused too
    assert t is None and "unrecorded" in why
    assert bench.traffic_from_profiles("k<3>", 10, "bbbbbbbbbbbbbbbb", str(p)) == (None, None)
    # the committed file: whatever it holds for the current build is quoted, anything else is not
    from mono_dataset_code_amd import capi

    for e in json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).values():
        t, why = bench.traffic_from_profiles(e["kernel"], 1, capi.code_id())
        assert (t is not None) == any(x.get("kernel") == e["kernel"] and x.get("code_id") == capi.code_id()
                                      for x in json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).values())


def test_bench_utilities_are_not_in_the_product_abi():
    """The synthetic frame generator and the linear-stream yardstick live in libmdc_bench.so (include/mdc_bench.h); the
    product libraries neither export nor import them."""
    import subprocess

    from mono_dataset_code_amd import build, capi

    names = declared("mdc_bench.h", "mdcb_")
    assert sorted(capi.BENCH_SYMBOLS) == names
    out = subprocess.run(["nm", "-D", build.LIB_BENCH], stdout=subprocess.PIPE, text=True, check=True).stdout
    for n in names:
        assert (" T " + n) in out, n
    for lib in (build.LIB_HIP, build.LIB_HOST, build.LIB_MULTI):
        sym = subprocess.run(["nm", "-D", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
        assert "mdcb_" not in sym and "synth_frames" not in sym and "ceiling_mix" not in sym, lib
        needed = subprocess.run(["readelf", "-d", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
        assert "libmdc_bench" not in needed, lib


def test_libmdc_host_exports_header():
    from mono_dataset_code_amd import capi

    names = declared("mdc_host.h", "mdch_")
    assert sorted(capi.HOST_SYMBOLS) == names
    lib = capi.host_lib()
    for n in names:
        assert hasattr(lib, n), n


def test_headers_compile_as_plain_c_and_cxx(tmp_path):
    """include/mdc_hip.h and mdc_host.h are a C ABI: they must parse as C99 and as C++ with nothing but libc headers."""
    import subprocess

    src = tmp_path / "abi.c"
    src.write_text('#include "mdc_host.h"\nint main(void){ mdc_info i; mdc_fov_model m; (void)i; (void)m; return MDC_OK; }\n')
    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I" + inc, str(src)],
                ["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I" + inc, str(src)]):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout


def test_product_library_is_built_with_shipped_switches():
    """No tuning / debug / diagnosis switch (csrc/mdc_build_config.h) differs from its shipped value in the library next to
    the package; a diagnosis switch without MDC_DIAGNOSIS_BUILD does not even compile."""
    import subprocess

    from mono_dataset_code_amd import build, capi

    assert capi.build_flags() == ""
    cfg = os.path.join(ROOT, "mono_dataset_code_amd", "csrc", "mdc_build_config.h")
    for sw in ("MDC_EXP_SKIP_STORE=1", "MDC_EXP_FAKE_COMPUTE=2", "MDC_EXP_STRIP_NOSAMPLE=1"):
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", "-D" + sw, cfg], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode != 0 and "diagnosis switch" in r.stdout, (sw, r.stdout)
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", "-DMDC_DIAGNOSIS_BUILD=1", "-D" + sw, cfg], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
    # the debug variant (bounds checks) names itself
    import ctypes

    dbg = ctypes.CDLL(build.build_debug())
    dbg.mdc_build_flags.restype = ctypes.c_char_p
    assert dbg.mdc_build_flags().decode() == "MDC_DEBUG_BOUNDS=1"


def test_signatures_have_no_torch_types():
    for h in ("mdc_hip.h", "mdc_host.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        assert "torch" not in txt and "at::" not in txt and "Tensor" not in txt


def test_no_cpu_fallback_without_gpu(calib_dirs):
    """On a box without a HIP device: context creation reports MDC_ERR_NO_DEVICE and the
    C++ classes leave their output untouched (and complain) -- they never compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mono_dataset_code_amd import capi

    with pytest.raises(capi.MdcError) as e:
        capi.Context(0)
    assert e.value.code == capi.ERR_NO_DEVICE
    d = calib_dirs["small_explicit"]
    fov = capi.UndistorterFOV(os.path.join(d, "camera.txt"))
    assert fov.is_valid() and not fov.has_gpu()
    W, H, w, h = fov.dims()
    out = np.full(w * h, 123.0, np.float32)
    fov.undistort(np.zeros(W * H, np.uint8), out)
    assert np.all(out == 123.0)
    ph = capi.PhotometricUndistorter(os.path.join(d, "pcalib.txt"), os.path.join(d, "vignette.png"), W, H)
    out2 = np.full(W * H, 123.0, np.float32)
    ph.unmap(np.zeros(W * H, np.uint8), out2, 1, 1, 1)
    assert np.all(out2 == 123.0)


def test_product_does_not_reference_the_oracle():
    """Nothing under the package or include/ may import, link or name oracle/ (the only
    shared piece is the Eigen stand-in, which lives on the product side)."""
    bad = []
    for base in ("mono_dataset_code_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp")) or f == "Core":
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle[/.]|liboracle|libmdc_ref|orc_\w+\(|ref_\w+\(", txt) and "oracle/Makefile" not in txt.replace("oracle's", ""):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_every_environment_variable_is_documented():
    """One table of every MDC_* environment variable: a getenv / os.environ read in the product that the table
    does not name fails here."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = doc[doc.index("## Environment"):doc.index("## Build")]
    used = set()
    for base in ("mono_dataset_code_amd", "include"):
        for dp, dn, fs in os.walk(os.path.join(ROOT, base)):
            dn[:] = [d for d in dn if d not in ("build", "variants", "__pycache__")]
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    used |= set(re.findall(r'getenv\("(MDC_[A-Z0-9_]+)"\)', txt))
                    used |= set(re.findall(r'environ(?:\.get)?[\[(]"(MDC_[A-Z0-9_]+)"', txt))
    txt = open(os.path.join(ROOT, "bench.py")).read()
    used |= set(re.findall(r'environ(?:\.get)?[\[(]"(MDC_[A-Z0-9_]+)"', txt))
    assert len(used) >= 15, used
    missing = sorted(v for v in used if ("`%s" % v) not in table)
    assert not missing, "undocumented environment variables: %s" % missing
