"""Fixed-QP layer scaling (svt_hip_vp9_layer_qindex) against the reference's own eb_vp9_compute_qdelta and delta_rate tables: the
committed fixture holds the reference's base_qindex for every (tune, hierarchical levels, temporal layer, qp); with oracle/_ref
present the fixture itself is re-derived from the reference (Codec/EbRateControlProcess.c:4680-4722)."""
import os
import subprocess

import numpy as np
import pytest

import svt_testlib as T

B = T.B
GOLD = os.path.join(T.GOLDEN_DIR, "qp_scaling_reference.npz")


def _check(rows):
    lib = B.load()
    assert len(rows) == 3 * 2 * 6 * 64
    for tune, levels, layer, qp, want in rows.tolist():
        assert lib.svt_hip_vp9_layer_qindex(qp, tune, levels, layer, 0) == want, (tune, levels, layer, qp, want)
    # a layer's q index never exceeds the sequence's, and the deepest layers are coded at the sequence q index
    for qp in range(64):
        base = lib.svt_hip_vp9_qindex_from_qp(qp)
        assert lib.svt_hip_vp9_layer_qindex(qp, 1, 4, 4, 0) == base and lib.svt_hip_vp9_layer_qindex(qp, 1, 4, 0, 0) <= base
        assert lib.svt_hip_vp9_layer_qindex(qp, 1, 4, 0, 1) == base     # key frames: not scaled here (QP_SCALING_MODE_1 is rate control)


def test_layer_qindex_vs_golden():
    _check(np.load(GOLD)["rows"])


@pytest.mark.skipif(not T.have_ref("ref_qp_scaling"), reason="oracle/_ref/ref_qp_scaling not built (reference absent)")
def test_layer_qindex_vs_reference_live():
    out = subprocess.check_output([os.path.join(T.REF_DIR, "ref_qp_scaling")]).decode()
    _check(np.array([[int(x) for x in line.split()] for line in out.strip().splitlines()], np.int16))
