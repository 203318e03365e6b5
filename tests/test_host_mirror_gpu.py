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
           "-L/opt/rocm/lib", "-lamdhip64", "-pthread", "-o", out]
    subprocess.check_call(cmd)
    return out


def test_host_mirror_compiles():
    """CPU: the header-only mirror and its test program compile and link against the C ABI."""
    assert os.path.exists(build_host_test())


@pytest.mark.gpu
def test_host_mirror_integrity_on_gpu(tmp_path):
    import sys

    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import flac_parse

    exe = build_host_test()
    res = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600)
    print(res.stdout[-4000:], res.stderr[-2000:])
    assert res.returncode == 0
    assert "all integrity tests passed" in res.stdout
    # the .flac the C++ mirror wrote (Stream::to_bytes: every frame packed by the GPU) decodes to its input
    data = open(os.path.join(str(tmp_path), "mirror.flac"), "rb").read()
    pcm = np.fromfile(os.path.join(str(tmp_path), "mirror.pcm"), np.int32).reshape(-1, 2)
    assert data[:4] == b"fLaC" and data[4] == 0x80
    si = data[8:42]
    assert int.from_bytes(si[0:2], "big") == 4096 and int.from_bytes(si[2:4], "big") == 4096
    packed = int.from_bytes(si[10:18], "big")
    assert packed >> 44 == 44100 and packed & ((1 << 36) - 1) == len(pcm)
    pos, t, number = 42, 0, 0
    while pos < len(data):
        got = flac_parse.parse_frame(data[pos:])
        assert got["number"] == number
        n = got["block_size"]
        assert np.array_equal(got["channels"], pcm[t:t + n].T), number
        pos, t, number = pos + got["length"], t + n, number + 1
    assert t == len(pcm) == 16123 and number == 4   # three full blocks + the 3835-sample tail
    for channels in (1, 3):   # mono / 3-channel streams: Independent(n) frames incl. an 808-sample tail
        data = open(os.path.join(str(tmp_path), f"mirror{channels}.flac"), "rb").read()
        pcm = np.fromfile(os.path.join(str(tmp_path), f"mirror{channels}.pcm"), np.int32).reshape(-1, channels)
        pos, t, number = 42, 0, 0
        while pos < len(data):
            got = flac_parse.parse_frame(data[pos:])
            assert got["number"] == number and got["channel_tag"] == channels - 1 and got["sample_rate"] == 48000
            n = got["block_size"]
            assert np.array_equal(got["channels"], pcm[t:t + n].T), (channels, number)
            pos, t, number = pos + got["length"], t + n, number + 1
        assert t == len(pcm) == 9000 and number == 3
