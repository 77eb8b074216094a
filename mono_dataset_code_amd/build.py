"""Build recipe for the native libraries (in-tree, no install step).

  libmdc_hip.so   hipcc, gfx950 only   HIP kernels + the C ABI (include/mdc_hip.h)
  libmdc_host.so  g++                  drop-in C++ classes + C facade (include/mdc_host.h)

Both land next to this file so they travel with the source tree to the GPU box.
`python -m mono_dataset_code_amd.build` rebuilds whatever is out of date.
"""
import os
import shutil
import subprocess
import time
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(CSRC, "host")
INC = os.path.join(ROOT, "include")
EIGEN_STUB = os.path.join(INC, "mono_dataset_code", "compat")  # only used when real Eigen is absent

LIB_HIP = os.path.join(PKG, "libmdc_hip.so")
LIB_HOST = os.path.join(PKG, "libmdc_host.so")

HIP_SOURCES = [os.path.join(CSRC, f) for f in ("mdc_kernels.hip", "mdc_vcal.hip", "mdc_jpeg.hip", "mdc_capi.hip", "mdc_plan.hip", "mdc_host_calls.hip",
                                                "mdc_pipeline.hip", "mdc_placement.hip")]
HIP_DEPS = HIP_SOURCES + [os.path.join(CSRC, "mdc_exports.map"), os.path.join(CSRC, "mdc_internal.h"), os.path.join(CSRC, "mdc_ctx.h"), os.path.join(CSRC, "mdc_build_config.h"), os.path.join(CSRC, "fov_point_model.h"), os.path.join(CSRC, "placement_classes.h"), os.path.join(INC, "mdc_hip.h")]
HOST_SOURCES = [os.path.join(HOST, f) for f in (
    "fov_undistorter.cpp", "photometric_undistorter.cpp", "gray_png.cpp", "host_device.cpp", "mdc_host_capi.cpp",
    "image_codecs.cpp", "image_codecs_ext.cpp", "zip_reader.cpp", "image_pool.cpp", "dataset_reader.cpp")]
HOST_DEPS = HOST_SOURCES + [os.path.join(HOST, "mdc_host_exports.map"), os.path.join(HOST, "gray_png.h"), os.path.join(HOST, "host_device.h"),
                            os.path.join(HOST, "image_codecs.h"), os.path.join(HOST, "image_codecs_internal.h"), os.path.join(HOST, "zip_reader.h"),
                            os.path.join(INC, "mono_dataset_code", "BenchmarkDatasetReader.h"),
                            os.path.join(INC, "mdc_hip.h"), os.path.join(INC, "mdc_host.h"),
                            os.path.join(INC, "mono_dataset_code", "FOVUndistorter.h"),
                            os.path.join(INC, "mono_dataset_code", "PhotometricUndistorter.h"),
                            os.path.join(INC, "mono_dataset_code", "ExposureImage.h"),
                            os.path.join(INC, "mono_dataset_code", "MdcBind.h")]

# -ffp-contract=off: the reference is built without FMA; contraction would break bit parity.
# -fvisibility=hidden: only what include/mdc_hip.h marks MDC_API is exported (tests/test_abi.py: exported set == header set).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
             "-fno-fast-math", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wall", "-Wno-unused-function"]
HOST_FLAGS = ["-O2", "-std=c++11", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wall"]
HOST_EXPORT_MAP = os.path.join(HOST, "mdc_host_exports.map")
# what still leaks through -fvisibility=hidden (libstdc++ template instantiations are declared with default visibility): made local
EXPORT_MAP = os.path.join(CSRC, "mdc_exports.map")
RECIPE = os.path.abspath(__file__)  # (the flags and mdc_code_id()'s rule live here: a changed recipe relinks, objects are redone by their own rule)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def eigen_include():
    """Real Eigen if installed, else the few-type stand-in shipped for this image."""
    for d in ("/usr/include/eigen3", "/usr/local/include/eigen3"):
        if os.path.exists(os.path.join(d, "Eigen", "Core")):
            return d
    return EIGEN_STUB


def code_text(src):
    """C / C++ source without its comments and blank space at line ends (string and character literals kept as they are):
    what the compiler sees of it, so that an edited comment does not make a measured build a different one."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c == "/" and i + 1 < n and src[i + 1] == "/":
            while i < n and src[i] != "\n":
                if src[i] == "\\" and i + 1 < n and src[i + 1] == "\n":  # a continued // comment
                    i += 1
                i += 1
        elif c == "/" and i + 1 < n and src[i + 1] == "*":
            j = src.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        elif c in "\"'":
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1])
            i = j + 1
        elif c in " \t\r\f\v":  # a run of blanks outside literals is one blank
            if not out or out[-1] != " ":
                out.append(" ")
            i += 1
        else:
            out.append(c)
            i += 1
    lines = [l.strip() for l in "".join(out).split("\n")]
    return "\n".join(l for l in lines if l)


def code_id(defines=()):
    """16 hex digits over everything the kernels are made of: sources and shared headers (comments and spacing aside), compile
    flags, -D list, compiler version."""
    import hashlib

    h = hashlib.sha256()
    for path in sorted(HIP_DEPS):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "r", encoding="utf-8", errors="surrogateescape") as f:
            h.update(code_text(f.read()).encode("utf-8", "surrogateescape"))
    h.update(" ".join(HIP_FLAGS + sorted(defines)).encode())
    try:
        h.update(subprocess.run([hipcc(), "--version"], stdout=subprocess.PIPE).stdout)
    except OSError:
        pass
    return h.hexdigest()[:16]


def _linked_from(lib, defines=()):
    """Does `lib` carry the identity of the sources as they are NOW?  (Modification times alone are fooled by an edit that lands while a
    build is running: the objects are from before it, the library's time stamp from after.)"""
    try:
        with open(lib, "rb") as f:
            return code_id(defines).encode() in f.read()
    except OSError:
        return False


def _compile_link_hip(out, defines=(), objdir_tag="product"):
    """Every .hip translation unit -> its own object, in parallel (the kernels TU alone takes a minute), then one link.
    An object is redone when its source, a shared header or the -D list changed."""
    from concurrent.futures import ThreadPoolExecutor

    started = time.time()
    identity = code_id(defines)  # of the sources as the compilers are about to read them
    objdir = os.path.join(PKG, "build", objdir_tag)
    os.makedirs(objdir, exist_ok=True)
    flags = [f for f in HIP_FLAGS if f != "-shared"] + ["-I" + INC] + ["-D" + d for d in defines]
    stamp = os.path.join(objdir, "flags.txt")
    flags_changed = not os.path.exists(stamp) or open(stamp).read() != " ".join(flags)
    headers = [d for d in HIP_DEPS if d not in HIP_SOURCES]
    jobs = []
    for src in HIP_SOURCES:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if flags_changed or _stale(obj, [src] + headers) or not os.path.exists(out) or not _linked_from(out, defines):
            jobs.append([hipcc()] + flags + ["-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as ex:
        list(ex.map(_run, jobs))
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    objs = [os.path.join(objdir, os.path.basename(src) + ".o") for src in HIP_SOURCES]
    # mdc_code_id(): the identity of this build (include/mdc_hip.h), a generated one-line translation unit
    idsrc = os.path.join(objdir, "mdc_code_id.cpp")
    with open(idsrc, "w") as f:
        f.write('#include "mdc_hip.h"\nextern "C" const char* mdc_code_id(void) { return "%s"; }\n' % identity)
    idobj = idsrc + ".o"
    _run(["g++", "-O1", "-fPIC", "-fvisibility=hidden", "-I" + INC, "-c", idsrc, "-o", idobj])
    _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + EXPORT_MAP] + objs + [idobj, "-o", out])
    if any(os.path.getmtime(d) > started for d in HIP_DEPS):
        sys.stderr.write("mono_dataset_code_amd.build: a source changed while %s was being built; it is stamped %s and the next build redoes it\n"
                         % (os.path.basename(out), identity))


def build_hip(force=False):
    if force or _stale(LIB_HIP, HIP_DEPS + [RECIPE]) or not _linked_from(LIB_HIP):
        if force:
            shutil.rmtree(os.path.join(PKG, "build", "product"), ignore_errors=True)
        _compile_link_hip(LIB_HIP)
    return LIB_HIP


def build_host(force=False):
    build_hip(force)
    if force or _stale(LIB_HOST, HOST_DEPS + [LIB_HIP]):
        _run(["g++"] + HOST_FLAGS + ["-I" + INC, "-I" + os.path.join(INC, "mono_dataset_code"), "-I" + HOST,
                                     "-I" + eigen_include()] + HOST_SOURCES +
             ["-L" + PKG, "-lmdc_hip", "-Wl,-rpath,$ORIGIN", "-Wl,--version-script=" + HOST_EXPORT_MAP, "-lz", "-lpthread", "-ldl", "-o", LIB_HOST])
    return LIB_HOST


LIB_BENCH = os.path.join(PKG, "libmdc_bench.so")
BENCH_SOURCE = os.path.join(CSRC, "bench", "mdc_bench.hip")


def build_bench(force=False):
    """libmdc_bench.so: measurement / test utilities (include/mdc_bench.h) -- not part of the product, not linked by it."""
    if force or _stale(LIB_BENCH, [BENCH_SOURCE, os.path.join(INC, "mdc_bench.h"), EXPORT_MAP]):
        _run([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-I" + INC, BENCH_SOURCE,
              "-Wl,--version-script=" + EXPORT_MAP, "-o", LIB_BENCH])
    return LIB_BENCH


LIB_MULTI = os.path.join(PKG, "libmdc_multi.so")
MULTI_SOURCE = os.path.join(CSRC, "mdc_multi.hip")


def build_multi(force=False):
    """libmdc_multi.so: one process, N GPUs, RCCL table broadcast (include/mdc_multi.h)."""
    build_hip(force)
    if force or _stale(LIB_MULTI, [MULTI_SOURCE, os.path.join(INC, "mdc_multi.h"), os.path.join(INC, "mdc_hip.h"), LIB_HIP, EXPORT_MAP]):
        rocm = os.path.dirname(os.path.dirname(os.path.realpath(hipcc())))
        _run([hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-I" + INC, MULTI_SOURCE,
              "-Wl,--version-script=" + EXPORT_MAP,
              "-L" + PKG, "-lmdc_hip", "-L" + os.path.join(rocm, "lib"), "-lrccl", "-lpthread",
              "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", LIB_MULTI])
    return LIB_MULTI


def build_host_sanitize(out):
    """tests/native/host_sanitize.cpp + the host sources under AddressSanitizer and UndefinedBehaviorSanitizer
    (tests/test_sanitize.py).  The HIP side is the normal libmdc_hip.so, not instrumented."""
    build_hip()
    src = [os.path.join(ROOT, "tests", "native", "host_sanitize.cpp")] + HOST_SOURCES
    _run(["g++", "-O1", "-g", "-std=c++11", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
          "-fno-omit-frame-pointer", "-Wall", "-I" + INC, "-I" + os.path.join(INC, "mono_dataset_code"), "-I" + HOST,
          "-I" + eigen_include()] + src + ["-L" + PKG, "-lmdc_hip", "-Wl,-rpath," + PKG, "-lz", "-lpthread", "-ldl", "-o", out])
    return out


def build_host_tsan(out):
    """tests/native/host_tsan.cpp + the host sources under ThreadSanitizer (tests/test_sanitize.py): the reader's decode pool,
    several readers, the image pool and the decoders from many threads.  The HIP side is the normal libmdc_hip.so."""
    build_hip()
    src = [os.path.join(ROOT, "tests", "native", "host_tsan.cpp")] + HOST_SOURCES
    _run(["g++", "-O1", "-g", "-std=c++11", "-ffp-contract=off", "-fsanitize=thread", "-fno-omit-frame-pointer", "-Wall", "-I" + INC,
          "-I" + os.path.join(INC, "mono_dataset_code"), "-I" + HOST, "-I" + eigen_include()] + src +
         ["-L" + PKG, "-lmdc_hip", "-Wl,-rpath," + PKG, "-lz", "-lpthread", "-ldl", "-o", out])
    return out


def build_variant(name, defines):
    """Experimental build of libmdc_hip with -D switches (see MDC_EXP_* in mdc_kernels.hip);
    loaded by tools/sweep.py --lib.  Lands in mono_dataset_code_amd/variants/."""
    d = os.path.join(PKG, "variants")
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, "libmdc_hip_%s.so" % name)
    diag = [] if name == "debug" else ["MDC_DIAGNOSIS_BUILD=1"]
    if _stale(out, HIP_DEPS + [RECIPE]) or not _linked_from(out, diag + list(defines)):
        # MDC_DIAGNOSIS_BUILD: the licence for the wrong-result switches of csrc/mdc_build_config.h -- only ever set here,
        # for a library that lands under variants/ and reports itself through mdc_build_flags()
        _compile_link_hip(out, diag + list(defines), objdir_tag="variant_" + name)
    return out


def build_debug():
    """The bounds-checking build of the kernels (tests/test_gpu_debug.py); travels to the GPU box like the product build."""
    return build_variant("debug", ["MDC_DEBUG_BOUNDS=1"])


def build_fault_injection():
    """libmdc_hip with MDC_EXP_HUFF_BAD_PROVISIONAL: the split Huffman kernel's segments publish wrong provisional exit states, so every
    right neighbour takes the path that relaxes a third time from the final state (tests/test_reader.py); right results, slower."""
    return build_variant("badprov", ["MDC_EXP_HUFF_BAD_PROVISIONAL=1"])


def build_all(force=False):
    build_hip(force)
    build_host(force)
    build_multi(force)
    build_bench(force)
    build_debug()
    build_fault_injection()
    return LIB_HIP, LIB_HOST, LIB_MULTI


if __name__ == "__main__":
    print("\n".join(build_all(force="--force" in sys.argv)))
