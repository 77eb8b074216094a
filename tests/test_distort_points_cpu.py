"""The device's distortCoordinates arithmetic, pinned on the CPU of the build box.

The kernel behind UndistorterFOV::distortCoordinates for bulk callers (reference src/main_vignetteCalib.cpp:284: 10^6 points
per image) must give the HOST libm's bits, so it restates glibc's fdlibm atanf instead of calling the GPU math library.
That restatement lives in a header both compilers take (mono_dataset_code_amd/csrc/fov_point_model.h); here the host
compiler builds it with the product's flags and compares it with this box's libm and with the class's own host loop
(tests/native/distort_points_cpu.cpp).  The GPU half -- the kernel against the host, and the class method routed to the
device for n >= 65536 -- is tests/test_gpu_parity.py::test_distort_*."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_point_model_header_equals_host_libm_and_the_class(calib_dirs, tmp_path):
    from mono_dataset_code_amd import build

    build.build_host()
    exe = str(tmp_path / "distort_points_cpu")
    inc = os.path.join(ROOT, "include")
    cmd = ["g++", "-O2", "-std=c++11", "-ffp-contract=off", "-Wall", "-I" + inc, "-I" + os.path.join(inc, "mono_dataset_code"),
           "-I" + os.path.join(ROOT, "mono_dataset_code_amd", "csrc"), "-I" + build.eigen_include(),
           os.path.join(ROOT, "tests", "native", "distort_points_cpu.cpp"), "-L" + build.PKG, "-lmdc_host", "-lmdc_hip",
           "-Wl,-rpath," + build.PKG, "-lm", "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    cams = [os.path.join(calib_dirs[k], "camera.txt") for k in sorted(calib_dirs)]
    env = dict(os.environ, MDC_DISTORT_GPU_MIN="0")  # the class's host loop is the thing compared with, GPU or not
    r = subprocess.run([exe] + cams, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=900)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and tail.startswith("ok "), r.stdout[-3000:]
    assert int(tail.split()[1]) > 50_000_000
