"""GPU parity against the REFERENCE's own outputs, not via the oracle: the committed fixtures (tests/golden/*.npz, produced by
running the reference built from source, tests/gen_golden.py) are compared directly with what the HIP kernels return through
the C ABI.  (The HIP transforms and the oracle's share the butterfly formulation -- this is the check that does not.)"""
import ctypes as C

import numpy as np
import pytest

import me_configs as MC
import svt_testlib as T
from test_gpu_me import hip_me_picture

B = T.B


def _golden(name):
    return dict(np.load(f"{T.GOLDEN_DIR}/{name}"))


@pytest.fixture(scope="module")
def ctx():
    lib = B.load()
    c = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
    yield c
    lib.svt_hip_ctx_destroy(c)


@pytest.mark.gpu
def test_me_hip_matches_reference_golden(ctx):
    g = _golden("me_reference.npz")
    assert len(g) == 9
    for key, ref in g.items():
        name, nl, tl, clip = key.split("|")
        nl, tl = int(nl), int(tl)
        gen = T.gen_clip if clip == "int" else T.gen_clip_subpel
        pics = [T.PaPic(f) for f in gen(264, 200, 3, 11)]
        res, _ = hip_me_picture(ctx, pics[1], pics[0], pics[2] if nl == 2 else None, MC.preset(name, nl, tl))
        assert not T.me_results_equal(ref, res, nl), key


@pytest.mark.gpu
def test_tq_hip_matches_reference_golden(ctx):
    g = _golden("tq_reference.npz")
    for seed in (1, 2):
        recon, q, dq, eob = T.hip_tq_batch(ctx, T.make_tq_case(seed))
        assert np.array_equal(q, g[f"q{seed}"]) and np.array_equal(dq, g[f"dq{seed}"])
        assert np.array_equal(eob, g[f"eob{seed}"]) and np.array_equal(recon, g[f"recon{seed}"])


@pytest.mark.gpu
def test_lf_hip_matches_reference_golden(ctx):
    g = _golden("lf_reference.npz")
    keys = sorted({k.split("|", 1)[1] for k in g})
    assert len(keys) == 3
    for k in keys:
        w, h, seed, sharp = (int(v) for v in k.split("|"))
        y, u, v = T.hip_lf_frame(ctx, T.make_lf_case(seed, w, h, sharp))
        assert np.array_equal(y, g["y|" + k]) and np.array_equal(u, g["u|" + k]) and np.array_equal(v, g["v|" + k])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sad_loop_hip_matches_reference_golden_and_oracle(ctx, seed):
    """svt_hip_sad_loop_batch_device = eb_vp9_sad_loop_kernel batched (row M1): HME-shaped jobs, tie-heavy windows, the
    largest level-0 window (224 x 112), a 1 x 1 window, planted exact matches."""
    case = T.make_sad_loop_case(seed)
    got = T.hip_sad_loop_case(ctx, case)
    assert np.array_equal(got, T.oracle_sad_loop_case(case))
    g = _golden("sad_loop_reference.npz")
    if str(seed) in g:
        assert np.array_equal(got, g[str(seed)])


@pytest.mark.parametrize("seed", [1, 2])
def test_sad_loop_oracle_matches_reference_golden(seed):
    case = T.make_sad_loop_case(seed)
    o = T.oracle_sad_loop_case(case)
    assert np.array_equal(o, _golden("sad_loop_reference.npz")[str(seed)])
    assert len(o) == 48 and (o[:, 0] == 0).sum() >= 5      # the planted exact matches are found


@pytest.mark.skipif(not T.have_ref("libsvtref_kernels.so"), reason="oracle/_ref not built (reference sources absent)")
def test_sad_loop_oracle_matches_reference_live():
    case = T.make_sad_loop_case(7)
    assert np.array_equal(T.oracle_sad_loop_case(case), T.ref_sad_loop_case(case))
