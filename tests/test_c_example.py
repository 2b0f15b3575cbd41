"""The C ABI from plain C (examples/me_from_c.c): compiles as C11 against include/svtvp9_hip.h and links the product
library; on a GPU it finds the motion it was given, without one it reports the missing device and exits non-zero."""
import os
import subprocess
import tempfile

import pytest

import svt_testlib as T


def _build(td):
    exe = os.path.join(td, "me_from_c")
    lib_dir = os.path.join(T.ROOT, "svt-vp9_amd")
    T.B.load()  # the library must have been built
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I", os.path.join(T.ROOT, "include"),
                           os.path.join(T.ROOT, "examples", "me_from_c.c"), "-L", lib_dir, "-lsvtvp9_hip", f"-Wl,-rpath,{lib_dir}", "-o", exe])
    return exe


def test_c_example_builds_and_fails_loudly_without_gpu():
    import torch
    with tempfile.TemporaryDirectory() as td:
        exe = _build(td)
        if torch.cuda.is_available():
            pytest.skip("GPU present: covered by the gpu test")
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 3 and "no usable GPU" in r.stderr


@pytest.mark.gpu
def test_c_example_finds_the_motion_on_gpu():
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([_build(td)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "partitions on the true motion" in r.stdout
