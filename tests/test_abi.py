"""CPU: the C-ABI library loads and exports every symbol include/svtvp9_hip.h declares; host-side (non-GPU) logic."""
import ctypes as C
import re

import numpy as np
import pytest

import svt_testlib as T

B = T.B


def test_library_exports_every_declared_symbol():
    lib = B.load()
    header = open(f"{T.ROOT}/include/svtvp9_hip.h").read()
    declared = set(re.findall(r"\b(svt_(?:hip|ivf)_[a-z0-9_]+)\s*\(", header))
    assert declared == set(B.EXPORTS), declared ^ set(B.EXPORTS)
    for s in declared:
        assert hasattr(lib, s), s


def test_struct_sizes_match_header():
    assert C.sizeof(B.Plane) == 32 and C.sizeof(B.PaPicture) == 96
    assert B.ME_RESULT_DTYPE.itemsize == 40 and B.TQ_BLOCK_DTYPE.itemsize == 32
    assert B.LF_MASK_DTYPE.itemsize == 160 and C.sizeof(B.LfThresh) == 192 and B.QUANT_DTYPE.itemsize == 20
    assert B.MC_MODE_INFO_DTYPE.itemsize == 12 and B.RATE_BLOCK_DTYPE.itemsize == 16 and B.RATE_TABLES_DTYPE.itemsize == 56472
    # the same sizes as the C compiler sees them
    import subprocess, tempfile, os
    src = ('#include <stdio.h>\n#include "svtvp9_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu",sizeof(svt_plane),sizeof(svt_me_pu_result),'
           'sizeof(svt_tq_block),sizeof(svt_lf_mask),sizeof(svt_mc_mode_info),sizeof(svt_rate_block),sizeof(svt_rate_tables),sizeof(svt_mc_picture));return 0;}')
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", f"{T.ROOT}/include", os.path.join(td, "s.c"), "-o", os.path.join(td, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(td, "s")]).split()]
    assert sizes == [32, 40, 32, 160, 12, 16, 56472, C.sizeof(B.McPicture)], sizes


def test_compute_entry_points_fail_loudly_without_gpu():
    """No CPU fallback: without a usable device the context cannot even be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = B.load().svt_hip_ctx_create(C.byref(ctx), 0)
    assert rc != 0 and not ctx.value
