"""svt_hip_minigop_split (host C: how a group shorter than a mini-GOP is cut into the units that get a prediction structure) against the
reference's own eb_vp9_generate_picture_window_split + eb_vp9_handle_incomplete_picture_window_map (golden fixture produced by
oracle/_ref/ref_pd_split, and live when that binary is present).  The low-delay P structure the reference then gives the parts that
are not a whole period (Codec/EbPictureDecisionProcess.c:1711-1727 + the tables of Codec/EbPredictionStructure.c) is picture
decision -- control plane, out of scope -- and is NOT reproduced: the encoder shim codes those parts as a chain of P pictures."""
import os

import numpy as np
import pytest

import svt_testlib as T


def _check(rows):
    for row in rows:
        n = int(row[0])
        want = [tuple(int(v) for v in row[1 + 3 * k:4 + 3 * k]) for k in range(4) if row[1 + 3 * k] >= 0]
        got = T.product_minigop_split(n)
        assert [g[:3] for g in got] == want, (n, got, want)
        for (start, length, lv, ra) in got:
            assert ra == int(length == (1 << lv))
        # released by an intra refresh (the intra picture is the group's last element): only the LAST part -- the one that ends
        # with the intra picture, mini_gop_idr_count > 0 -- is forced to low-delay P; earlier whole periods stay random access
        cut = T.product_minigop_split(n, 4, 1)
        assert [g[:3] for g in cut] == want and cut[-1][3] == 0
        assert all(g[3] == int(g[1] == (1 << g[2])) for g in cut[:-1])


def test_split_matches_reference_golden():
    _check(np.load(os.path.join(T.GOLDEN_DIR, "pd_split_reference.npz"))["split"])
    assert T.product_minigop_split(1) == [(0, 1, 4, 0)]
    assert T.product_minigop_split(8, 3) == [(0, 8, 3, 1)] and T.product_minigop_split(5, 3) == [(0, 5, 3, 0)]


@pytest.mark.skipif(not T.have_ref("ref_pd_split"), reason="oracle/_ref/ref_pd_split not built")
def test_split_matches_reference_live():
    _check(T.ref_minigop_split())
