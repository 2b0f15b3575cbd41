"""GPU parity: HIP inter prediction (8-tap motion compensation, through the C ABI) vs the oracle and the golden
fixture produced by the reference's own inter_prediction(), bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import svt_testlib as T
from gen_golden import MC_GOLDEN_CASES

B = T.B
pytestmark = pytest.mark.gpu
GOLD = os.path.join(T.GOLDEN_DIR, "mc_reference.npz")


@pytest.fixture(scope="module")
def ctx():
    lib = B.load()
    c = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
    yield c
    lib.svt_hip_ctx_destroy(c)


def _same(a, b):
    for name, x, y in zip("yuv", a, b):
        assert np.array_equal(x, y), (name, int(np.sum(x != y)), np.argwhere(x != y)[:6].tolist())


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("use_subpel", [1, 0])
def test_mc_vs_oracle(ctx, seed, use_subpel):
    case = T.make_mc_case(seed, use_subpel=use_subpel)
    _same(T.hip_mc_frame(ctx, case), T.oracle_mc_frame(case))


@pytest.mark.parametrize("w,h", [(64, 64), (136, 72), (264, 200), (520, 264)])
def test_mc_ragged_sizes_and_far_mvs(ctx, w, h):
    """pictures that are not multiples of 64 (partial superblocks, partial 32-unit workgroups) and MVs that the
    reference clamps to the picture border"""
    case = T.make_mc_case(31 + w, width=w, height=h, mv_range=600)
    _same(T.hip_mc_frame(ctx, case), T.oracle_mc_frame(case))


@pytest.mark.parametrize("seed,w,h,sub", MC_GOLDEN_CASES)
def test_mc_vs_reference_golden(ctx, seed, w, h, sub):
    g = np.load(GOLD)
    case = T.make_mc_case(seed, width=w, height=h, use_subpel=sub)
    got = T.hip_mc_frame(ctx, case)
    my, mc = T.mc_inter_masks(case)
    for p, a, m in zip("yuv", got, (my, mc, mc)):
        assert np.array_equal(a[m], g[f"{p}|{seed}|{w}|{h}|{sub}"][m]), p
        assert (a[~m] == 0x5A).all()


def test_mc_all_phases_every_block_size(ctx):
    """every (x, y) filter phase pair on every square block size: 256 phase pairs spread over the blocks"""
    case = T.make_mc_case(77, width=512, height=256, rect=False, intra_share=0.0)
    mi = case["mi"]
    k = 0
    for r in range(case["mi_rows"]):
        for c in range(case["mi_cols"]):
            if r % mi[r, c]["bh8"] == 0 and c % mi[r, c]["bw8"] == 0:
                h8, w8 = int(mi[r, c]["bh8"]), int(mi[r, c]["bw8"])
                mi[r:r + h8, c:c + w8]["mv_row"] = (((k >> 4) & 15) - 40, (k & 15) + 24)       # 1/8 sample: 16 phases each ...
                mi[r:r + h8, c:c + w8]["mv_col"] = ((k & 15) + 16, ((k >> 4) & 15) - 56)
                k += 1
    _same(T.hip_mc_frame(ctx, case), T.oracle_mc_frame(case))


def test_mc_batch_device_two_pictures(ctx):
    """svt_hip_inter_pred_batch_device: two pictures of different sizes in one launch, device pointers"""
    import torch
    lib = B.load()
    cases = [T.make_mc_case(41, width=192, height=128), T.make_mc_case(42, width=136, height=72, use_subpel=0)]
    keep, pics = [], (B.McPicture * 2)()

    def dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()
        keep.append(t)
        return t

    outs = []
    for i, case in enumerate(cases):
        W, H, pad = case["width"], case["height"], case["pad"]
        pics[i].d_mi = dev(case["mi"]).data_ptr()
        pics[i].mi_stride, pics[i].mi_rows, pics[i].mi_cols, pics[i].use_subpel = case["mi_cols"], case["mi_rows"], case["mi_cols"], case["use_subpel"]
        for l, (y, u, v) in enumerate(case["refs"]):
            ty, tu, tv = dev(y), dev(u), dev(v)
            r = pics[i].ref[l]
            r.y = ty.data_ptr() + pad * y.shape[1] + pad
            r.u = tu.data_ptr() + (pad // 2) * u.shape[1] + pad // 2
            r.v = tv.data_ptr() + (pad // 2) * v.shape[1] + pad // 2
            r.y_stride, r.uv_stride, r.width, r.height = y.shape[1], u.shape[1], W, H
        o = [dev(np.full((H, W), 0x5A, np.uint8)), dev(np.full((H // 2, W // 2), 0x5A, np.uint8)), dev(np.full((H // 2, W // 2), 0x5A, np.uint8))]
        outs.append(o)
        p = pics[i].pred
        p.y, p.u, p.v, p.y_stride, p.uv_stride, p.width, p.height = o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), W, W // 2, W, H
    B.check(lib.svt_hip_inter_pred_batch_device(ctx, 2, pics))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    for case, o in zip(cases, outs):
        W, H = case["width"], case["height"]
        got = [o[0].cpu().numpy().reshape(H, W), o[1].cpu().numpy().reshape(H // 2, W // 2), o[2].cpu().numpy().reshape(H // 2, W // 2)]
        _same(got, T.oracle_mc_frame(case))
