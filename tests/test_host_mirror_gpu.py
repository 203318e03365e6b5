"""Builds and runs the C++ host-mirror integrity test (tests/host/host_mirror_test.cpp): the
reference's e2e test (src/lib.rs:201-251) on `flacenc::encode_with_fixed_block_size` with every
LPC candidate computed by the GPU through the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_host_test():
    out = os.path.join(ROOT, "tests", "host", "host_mirror_test")
    cmd = ["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "flacenc_rs_amd", "host"),
           os.path.join(ROOT, "tests", "host", "host_mirror_test.cpp"),
           "-L", os.path.join(ROOT, "flacenc_rs_amd"), "-lflacenc_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "flacenc_rs_amd"), "-Wl,-rpath,/opt/rocm/lib",
           "-L/opt/rocm/lib", "-lamdhip64", "-o", out]
    subprocess.check_call(cmd)
    return out


def test_host_mirror_compiles():
    """CPU: the header-only mirror and its test program compile and link against the C ABI."""
    assert os.path.exists(build_host_test())


@pytest.mark.gpu
def test_host_mirror_integrity_on_gpu():
    exe = build_host_test()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(res.stdout[-4000:], res.stderr[-2000:])
    assert res.returncode == 0
    assert "all integrity tests passed" in res.stdout
