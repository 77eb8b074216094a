"""The buffer-placement allocator's host arithmetic (mono_dataset_code_amd/csrc/placement_classes.h: the split of probe times into memory
classes, the classes of the groups from two references, which pieces make up which range) under g++ on synthetic times -- no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_placement_arithmetic_on_synthetic_times(tmp_path):
    exe = str(tmp_path / "placement_classes_cpu")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "mono_dataset_code_amd", "csrc"),
                        os.path.join(ROOT, "tests", "native", "placement_classes_cpu.cpp"), "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "FAIL" not in r.stdout, r.stdout
    assert r.stdout.count(" ok") >= 13, r.stdout
