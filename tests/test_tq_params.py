"""Host-side quantiser tables (svt_hip_quant_tables_init) against the reference's eb_vp9_init_quantizer: the committed
fixture holds its output for all 256 q indices; with oracle/_ref present the fixture itself is re-derived from the reference."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import svt_testlib as T
from gen_golden import QUANT_GOLDEN_DELTAS

B = T.B
GOLD = os.path.join(T.GOLDEN_DIR, "quant_reference.npz")


def _check(rows):
    """row: q, y_dc_step(delta 0), then for Y and UV: dc_step ac_step + (zbin round quant quant_shift dequant) x [DC, AC]"""
    lib = B.load()
    for r in rows:
        q, ydc = int(r[0]), int(r[1])
        for base in (2, 14):
            t = np.zeros(1, dtype=B.QUANT_DTYPE)
            assert lib.svt_hip_quant_tables_init(q, ydc, int(r[base]), int(r[base + 1]), t.ctypes.data_as(C.c_void_p)) == 0
            for i in range(2):
                want = r[base + 2 + 5 * i: base + 7 + 5 * i]
                got = [int(t[n][0][i]) for n in ("zbin", "round", "quant", "quant_shift", "dequant")]
                assert got == [int(x) for x in want], (q, base, i, got, want)


@pytest.mark.parametrize("deltas", QUANT_GOLDEN_DELTAS)
def test_quant_tables_vs_golden(deltas):
    _check(np.load(GOLD)["|".join(map(str, deltas))])


@pytest.mark.skipif(not T.have_ref("ref_quant_tables"), reason="oracle/_ref/ref_quant_tables not built (reference absent)")
def test_quant_tables_vs_reference_live():
    out = subprocess.check_output([os.path.join(T.REF_DIR, "ref_quant_tables"), "2", "-7", "9"]).decode()
    _check(np.array([[int(x) for x in line.split()] for line in out.strip().splitlines()], np.int32))


def test_quant_tables_rejects_bad_arguments():
    t = np.zeros(1, dtype=B.QUANT_DTYPE)
    lib = B.load()
    assert lib.svt_hip_quant_tables_init(256, 8, 8, 8, t.ctypes.data_as(C.c_void_p)) != 0
    assert lib.svt_hip_quant_tables_init(10, 8, 0, 8, t.ctypes.data_as(C.c_void_p)) != 0
    assert lib.svt_hip_quant_tables_init(10, 8, 8, 8, None) != 0
