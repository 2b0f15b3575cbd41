"""GPU parity of the picture-level EncDec driver (svt_hip_encdec_batch_device and the stand-in decision) through the C ABI against
the oracle chain (tests/encdec_model.py): device-built block lists == host lists, prediction, coefficients, eob map, skip flags,
masks, deblocked and padded reconstruction -- all bit-exact, for every combination of the stage flags the reference derives."""
import ctypes as C

import numpy as np
import pytest
import torch

import encdec_model as M
import me_configs as MC
import svt_testlib as T

B = T.B
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    lib = B.load()
    c = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
    yield c
    lib.svt_hip_ctx_destroy(c)


def _chroma(y, k):
    u = (y[::2, ::2].astype(np.int32) // 2 + 32 + 8 * k).astype(np.uint8)
    v = (255 - y[::2, ::2] // 2 - (y[1::2, 1::2] // 4)).astype(np.uint8)
    return u, v


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


class DevPicture:
    """device buffers of one picture of a batch, carved out of the batch's slabs"""

    def __init__(self, W, H, src, refs_dev, mc_mi, lf_mi, slab_src, slab_pred, slab_q, slab_dq, index, rec_init):
        self.W, self.H = W, H
        pic = W * H * 3 // 2
        self.src_t = slab_src[index * pic:(index + 1) * pic]
        self.src_t.copy_(torch.from_numpy(np.concatenate([p.ravel() for p in src])))
        self.pred_t = slab_pred[index * pic:(index + 1) * pic]
        nco = T.n_sb(W, H) * B.SB_COEFFS
        self.q_t, self.dq_t = slab_q[index * nco:(index + 1) * nco], slab_dq[index * nco:(index + 1) * nco]
        self.rec = M.RefPic(W, H)
        self.rec_t = dev(rec_init.buf)
        self.mc_t, self.lf_t = dev(mc_mi.view(np.uint8)), dev(lf_mi.view(np.uint8))
        self.emap_t = torch.zeros(M.eob_map_offsets(W, H)[3], dtype=torch.int16, device="cuda")
        self.lfm_t = torch.zeros(T.n_sb(W, H) * 160, dtype=torch.uint8, device="cuda")
        self.nz_t = torch.full((lf_mi.size,), 7, dtype=torch.uint8, device="cuda")
        self.refs_dev = refs_dev

    def tight(self, base, stride_w):
        d = B.YuvPlanes()
        W, H = self.W, self.H
        d.y, d.u, d.v = base, base + W * H, base + W * H + (W // 2) * (H // 2)
        d.y_stride, d.uv_stride, d.width, d.height = W, W // 2, W, H
        return d

    def struct(self, refs, use_subpel=1, has_intra=0):
        p = B.EncdecPicture()
        p.has_intra = has_intra
        p.d_mc_mi, p.d_lf_mi = self.mc_t.data_ptr(), self.lf_t.data_ptr()
        p.src, p.pred = self.tight(self.src_t.data_ptr(), self.W), self.tight(self.pred_t.data_ptr(), self.W)
        p.recon = self.rec.desc(self.rec_t.data_ptr())
        for l in range(2):
            p.ref[l] = refs[l].desc(self.refs_dev[l].data_ptr())
        p.d_qcoeff, p.d_dqcoeff, p.d_eob_map = self.q_t.data_ptr(), self.dq_t.data_ptr(), self.emap_t.data_ptr()
        p.d_lfm, p.d_nz, p.use_subpel = self.lfm_t.data_ptr(), self.nz_t.data_ptr(), use_subpel
        return p


def make_inputs(W, H, n_pics, seed, preset="c2_1080p_m8"):
    frames = T.gen_clip_subpel(W, H, n_pics + 2, seed)
    refs = []
    for k in (0, n_pics + 1):
        refs.append(M.RefPic(W, H).set_padded(frames[k], *_chroma(frames[k], k)))
    pics = [T.PaPic(f) for f in frames]
    p = MC.preset(preset, 2, 1)
    me = [T.oracle_me_picture_mt(pics[i], pics[0], pics[n_pics + 1], p)[0] for i in range(1, n_pics + 1)]
    srcs = [(frames[i],) + _chroma(frames[i], i) for i in range(1, n_pics + 1)]
    return srcs, refs, me


def masks_equal(a, b):
    """field-wise (the 6 bytes of tail padding of a LOOP_FILTER_MASK carry nothing)"""
    return all(np.array_equal(a[f], b[f]) for f in B.LF_MASK_DTYPE.names)


def flags_of(**kw):
    c, o = B.EncdecFlagsConfig(**kw), B.EncdecFlags()
    assert B.load().svt_hip_encdec_flags_derive(C.byref(c), C.byref(o)) == 0
    return o


def md_host(me, W, H, lam, level):
    mi_rows, mi_cols = H // 8, W // 8
    mc = np.zeros((mi_rows, mi_cols), dtype=B.MC_MODE_INFO_DTYPE)
    lf = np.zeros((mi_rows, mi_cols), dtype=B.LF_MODE_INFO_DTYPE)
    assert B.load().svt_hip_md_default_picture(me.ctypes.data_as(C.c_void_p), W, H, lam, level, mc.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p), mi_cols) == 0
    return mc, lf


def run_device(ctx, W, H, srcs, refs, grids, q_index, flags, thr, rec_inits, has_intra=0, no_dq=False):
    lib = B.load()
    n = len(srcs)
    pic = W * H * 3 // 2
    nco = T.n_sb(W, H) * B.SB_COEFFS
    slab_src, slab_pred = torch.zeros(n * pic, dtype=torch.uint8, device="cuda"), torch.zeros(n * pic, dtype=torch.uint8, device="cuda")
    slab_q, slab_dq = torch.zeros(n * nco, dtype=torch.int16, device="cuda"), torch.zeros(n * nco, dtype=torch.int16, device="cuda")
    refs_dev = [dev(r.buf) for r in refs]
    dp = [DevPicture(W, H, srcs[i], refs_dev, grids[i][0], grids[i][1], slab_src, slab_pred, slab_q, slab_dq, i, rec_inits[i]) for i in range(n)]
    arr = (B.EncdecPicture * n)(*[d.struct(refs, has_intra=has_intra) for d in dp])
    if no_dq:   # svt_encdec_picture.d_dqcoeff = NULL: the dequantised coefficients never leave the lane
        for k in range(n):
            arr[k].d_dqcoeff = None
    work = C.c_void_p()
    B.check(lib.svt_hip_encdec_work_create(ctx, n, W, H, C.byref(work)))
    torch.cuda.synchronize()
    try:
        B.check(lib.svt_hip_encdec_batch_device(ctx, work, n, arr, W, H, W // 8, q_index, C.byref(flags), C.byref(thr), M.PAD, M.PAD))
        cnt = (C.c_int32 * 8)()
        B.check(lib.svt_hip_encdec_work_status(ctx, work, cnt))
        total = cnt[3] + cnt[7]
        blocks, pos, eob = np.zeros(total, dtype=B.TQ_BLOCK_DTYPE), np.zeros(total, np.uint32), np.zeros(total, np.uint16)
        got = lib.svt_hip_encdec_work_download(ctx, work, blocks.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p), eob.ctypes.data_as(C.c_void_p), total)
        assert got == total
    finally:
        lib.svt_hip_encdec_work_destroy(ctx, work)
    return dp, blocks, pos, eob, list(cnt)


@pytest.mark.parametrize("W,H,n_pics,q_index,cfg", [
    (256, 192, 2, 160, dict(enc_mode=8, tune=1, temporal_layer_index=0, is_used_as_reference=1, recon_file=0, loop_filter=1)),   # everything on
    (200, 136, 3, 208, dict(enc_mode=8, tune=1, temporal_layer_index=2, is_used_as_reference=1, recon_file=0, loop_filter=1)),   # no deblocking (mismatch allowed)
    (136, 72, 2, 100, dict(enc_mode=8, tune=1, temporal_layer_index=4, is_used_as_reference=0, recon_file=0, loop_filter=1)),    # no reconstruction at all
    (200, 136, 2, 236, dict(enc_mode=8, tune=1, temporal_layer_index=4, is_used_as_reference=0, recon_file=1, loop_filter=1)),   # recon output: filtered, not padded
    (136, 72, 11, 180, dict(enc_mode=8, tune=1, temporal_layer_index=0, is_used_as_reference=1, recon_file=0, loop_filter=1)),   # more pictures than reconstruction bases
])
@pytest.mark.parametrize("sb_order", [0, 1], ids=["size-grouped", "sb-ordered"])
def test_encdec_batch_vs_oracle_chain(ctx, W, H, n_pics, q_index, cfg, sb_order, monkeypatch):
    # both forms of the transform stage: four size-grouped launches (default) / SB-ordered lists + svt_tq_sb_kernel (SVT_HIP_TQ_SB_ORDER=1, read when
    # the driver's workspace is created)
    monkeypatch.setenv("SVT_HIP_TQ_SB_ORDER", str(sb_order))
    lib = B.load()
    srcs, refs, me = make_inputs(W, H, n_pics, seed=W + n_pics)
    level = lib.svt_hip_lf_level_from_q(lib.svt_hip_vp9_ac_step(q_index), 0)
    grids = [md_host(m, W, H, 300, level) for m in me]
    flags = flags_of(**cfg)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    rng = np.random.default_rng(5)
    rec_inits = [M.RefPic(W, H) for _ in range(n_pics)]
    for r in rec_inits:
        r.buf[:] = rng.integers(0, 256, r.buf.size, dtype=np.uint8)      # a recycled buffer holds junk
    dp, blocks, pos, eob, cnt = run_device(ctx, W, H, srcs, refs, grids, q_index, flags, thr, rec_inits)
    # the lists the device built == the host lists of the same grids (batch order: size, picture, SB, unit)
    # the driver's reconstruction bases: the pictures' lowest plane addresses in ascending order, a new base wherever a picture does not
    # lie within 4 GB above the last one (span = the largest stride x rows among the batch's planes)
    all_ptrs = [[dp[i].rec_t.data_ptr() + o for o in rec_inits[i].offsets()] for i in range(n_pics)]
    span = max(W, rec_inits[0].pw) * H
    set_base = []
    for ptrs in sorted(all_ptrs, key=min):
        if not set_base or max(ptrs) - set_base[-1] + span >= 2 ** 32:
            set_base.append(min(ptrs))
    geoms = []
    for i in range(n_pics):
        g = B.TqPicGeom()
        g.width, g.height = W, H
        for k, o in enumerate((0, W * H, W * H + (W // 2) * (H // 2))):
            g.src_off[k] = g.pred_off[k] = i * (W * H * 3 // 2) + o
        ptrs = all_ptrs[i]
        k_set = max(k for k, b in enumerate(set_base) if min(ptrs) >= b and max(ptrs) - b + span < 2 ** 32)
        for k, ptr in enumerate(ptrs):
            g.recon_off[k] = ptr - set_base[k_set]
        g.src_stride[0] = g.pred_stride[0] = W
        g.src_stride[1] = g.pred_stride[1] = W // 2
        g.recon_stride[0], g.recon_stride[1] = rec_inits[i].pw, rec_inits[i].cpw
        g.coeff_base, g.recon_set, g.pic, g.do_recon = i * T.n_sb(W, H) * B.SB_COEFFS, k_set, i, int(flags.do_recon)
        geoms.append(g)
    hb, hp, hc = M.host_block_list([g_[1] for g_ in grids], geoms, W // 8)
    assert [cnt[4 + s] for s in range(4)] == hc and cnt[:4] == [0, hc[0], hc[0] + hc[1], hc[0] + hc[1] + hc[2]]
    # the device's list holds the same blocks; its ORDER is the device's own business (round 6: [picture][chunk of SBs][size][SB][unit][plane] so that
    # one transform launch reads every SB once -- SVT_HIP_TQ_SB_ORDER=0 restores [size][picture][SB]): compared in the order of the position codes
    od, oh = np.argsort(pos, kind="stable"), np.argsort(hp, kind="stable")
    assert len(np.unique(pos)) == len(pos)
    assert np.array_equal(pos[od], hp[oh]) and blocks[od].tobytes() == hb[oh].tobytes()
    kinds = set()
    for i in range(n_pics):
        rec0 = M.RefPic(W, H)
        rec0.buf[:] = rec_inits[i].buf
        o = oracle = M.oracle_encdec_picture(srcs[i], refs, grids[i][0], grids[i][1], q_index, flags, thr, recon_init=rec0)
        d = dp[i]
        pred_g = d.pred_t.cpu().numpy()
        assert np.array_equal(pred_g, np.concatenate([p.ravel() for p in o["pred"]])), "prediction"
        assert np.array_equal(d.q_t.cpu().numpy(), o["qcoeff"]) and np.array_equal(d.dq_t.cpu().numpy(), o["dqcoeff"]), "coefficients"
        assert np.array_equal(d.emap_t.cpu().numpy().view(np.uint16), o["eob_map"]), "eob map"
        lf_g = d.lf_t.cpu().numpy().view(B.LF_MODE_INFO_DTYPE).reshape(H // 8, W // 8)
        assert np.array_equal(lf_g["skip"], o["lf_mi"]["skip"]), "skip flags"
        kinds |= set(lf_g["sb_type"].ravel().tolist())
        rec_g = d.rec_t.cpu().numpy()
        if flags.apply_loop_filter:
            lfm_g = d.lfm_t.cpu().numpy().view(B.LF_MASK_DTYPE).reshape(o["lfm"].shape)
            assert masks_equal(lfm_g, o["lfm"]), "loop-filter masks"
        assert np.array_equal(rec_g, o["rec"].buf), ("reconstruction", int(np.sum(rec_g != o["rec"].buf)))
        if not flags.do_recon:
            assert np.array_equal(rec_g, rec_inits[i].buf)               # untouched
        else:
            for a, b in zip(o["rec"].interior(rec_g), srcs[i]):
                assert np.mean(np.abs(a.astype(np.int32) - b)) < 16       # it IS a reconstruction of the source
    assert len(kinds) >= 2                                                # the stand-in decision really partitions


def test_encdec_batch_without_dqcoeff(ctx):
    """d_dqcoeff = NULL (the encode pass of the bench and of the encoder library): same quantised coefficients, same reconstruction,
    and the dqcoeff buffers are not written"""
    lib = B.load()
    W, H, n_pics, q_index = 200, 136, 3, 150
    srcs, refs, me = make_inputs(W, H, n_pics, seed=77)
    level = lib.svt_hip_lf_level_from_q(lib.svt_hip_vp9_ac_step(q_index), 0)
    grids = [md_host(m, W, H, 300, level) for m in me]
    flags = flags_of(enc_mode=8, tune=1, temporal_layer_index=0, is_used_as_reference=1, recon_file=0, loop_filter=1)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    out = []
    for no_dq in (False, True):
        rec_inits = [M.RefPic(W, H) for _ in range(n_pics)]
        grids_k = [(mc.copy(), lf.copy()) for mc, lf in grids]
        dp, blocks, pos, eob, cnt = run_device(ctx, W, H, srcs, refs, grids_k, q_index, flags, thr, rec_inits, no_dq=no_dq)
        out.append(([d.q_t.cpu().numpy() for d in dp], [d.dq_t.cpu().numpy() for d in dp], [d.rec_t.cpu().numpy() for d in dp], eob))
    for i in range(n_pics):
        assert np.array_equal(out[0][0][i], out[1][0][i]) and np.array_equal(out[0][2][i], out[1][2][i])
        assert out[0][1][i].any() and not out[1][1][i].any()
    assert np.array_equal(out[0][3], out[1][3])


def test_md_default_device_equals_host(ctx):
    lib = B.load()
    W, H = 200, 136
    srcs, refs, me = make_inputs(W, H, 3, seed=9)
    mi_n = (H // 8) * (W // 8)
    res_t = [dev(m.view(np.uint8)) for m in me]
    mc_t = [torch.zeros(mi_n * 12, dtype=torch.uint8, device="cuda") for _ in me]
    lf_t = [torch.full((mi_n * 8,), 9, dtype=torch.uint8, device="cuda") for _ in me]
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    torch.cuda.synchronize()
    B.check(lib.svt_hip_md_default_batch_device(ctx, 3, arr(res_t), W, H, 250, 21, arr(mc_t), arr(lf_t), W // 8))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    for i, m in enumerate(me):
        mc, lf = md_host(m, W, H, 250, 21)
        assert mc_t[i].cpu().numpy().tobytes() == mc.tobytes() and lf_t[i].cpu().numpy().tobytes() == lf.tobytes()


def test_lf_masks_device_equals_host(ctx):
    lib = B.load()
    mi_rows, mi_cols = 17, 25
    grids = [T.gen_mode_info_grid(s, mi_rows, mi_cols, mi_stride=mi_cols)[2] for s in (3, 4)]
    nsb = ((mi_rows + 7) // 8) * ((mi_cols + 7) // 8)
    g_t = [dev(g.view(np.uint8)) for g in grids]
    m_t = [torch.full((nsb * 160,), 0xEE, dtype=torch.uint8, device="cuda") for _ in grids]
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    torch.cuda.synchronize()
    B.check(lib.svt_hip_lf_build_masks_device(ctx, 2, arr(g_t), mi_cols, mi_rows, mi_cols, arr(m_t)))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    for g, m in zip(grids, m_t):
        want = T.product_lf_build_masks(g, mi_rows, mi_cols)
        assert masks_equal(m.cpu().numpy().view(B.LF_MASK_DTYPE).reshape(want.shape), want)


def test_malformed_grid_is_reported(ctx):
    lib = B.load()
    W, H = 136, 72
    srcs, refs, me = make_inputs(W, H, 1, seed=4)
    level = 10
    mc, lf = md_host(me[0], W, H, 300, level)
    lf["sb_type"][8, 0] = 12                                              # a 64x64 block in the 8-row bottom strip: crosses the edge
    flags = flags_of(enc_mode=8, tune=1, temporal_layer_index=0, is_used_as_reference=1, recon_file=0, loop_filter=1)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    with pytest.raises(RuntimeError):
        run_device(ctx, W, H, srcs, refs, [(mc, lf)], 160, flags, thr, [M.RefPic(W, H)])


@pytest.mark.parametrize("W,H,n_pics,q_index,cfg", [
    (256, 192, 2, 160, dict(enc_mode=8, tune=1, temporal_layer_index=0, is_used_as_reference=1, recon_file=0, loop_filter=1)),
    (200, 136, 3, 120, dict(enc_mode=8, tune=1, temporal_layer_index=2, is_used_as_reference=1, recon_file=0, loop_filter=1)),   # reconstructed, not deblocked
    (136, 72, 2, 200, dict(enc_mode=8, tune=1, temporal_layer_index=4, is_used_as_reference=0, recon_file=1, loop_filter=1)),
])
def test_intra_blocks_inside_inter_pictures(ctx, W, H, n_pics, q_index, cfg):
    """inter pictures with a share of intra blocks (random modes): the batch codes the inter blocks, the intra pass behind it the intra
    blocks from their neighbours' reconstruction; everything downstream (skip flags, masks, deblocking, border) sees both"""
    lib = B.load()
    srcs, refs, me = make_inputs(W, H, n_pics, seed=W + n_pics + 1)
    level = lib.svt_hip_lf_level_from_q(lib.svt_hip_vp9_ac_step(q_index), 0)
    grids, n_intra = [], 0
    for i, m in enumerate(me):
        mc, lf = md_host(m, W, H, 300, level)
        lf2, mc2, n = M.make_mixed(50 + i, lf, mc, share=0.3, level=level)
        grids.append((mc2, lf2))
        n_intra += n
    assert n_intra > 10
    flags = flags_of(**cfg)
    assert flags.do_recon
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    rng = np.random.default_rng(6)
    rec_inits = [M.RefPic(W, H) for _ in range(n_pics)]
    for r in rec_inits:
        r.buf[:] = rng.integers(0, 256, r.buf.size, dtype=np.uint8)
    dp, blocks, pos, eob, cnt = run_device(ctx, W, H, srcs, refs, grids, q_index, flags, thr, rec_inits, has_intra=1)
    for i in range(n_pics):
        rec0 = M.RefPic(W, H)
        rec0.buf[:] = rec_inits[i].buf
        o = M.oracle_encdec_picture(srcs[i], refs, grids[i][0], grids[i][1], q_index, flags, thr, recon_init=rec0)
        d = dp[i]
        inter_y = np.kron(grids[i][1]["is_inter"] == 1, np.ones((8, 8), bool))
        pred_g = d.pred_t.cpu().numpy()
        want_pred = np.concatenate([p.ravel() for p in o["pred"]])
        assert np.array_equal(pred_g, want_pred), "prediction (inter blocks from MC, intra blocks from the intra pass)"
        assert np.array_equal(d.q_t.cpu().numpy(), o["qcoeff"]) and np.array_equal(d.dq_t.cpu().numpy(), o["dqcoeff"]), "coefficients"
        assert np.array_equal(d.emap_t.cpu().numpy().view(np.uint16), o["eob_map"]), "eob map"
        lf_g = d.lf_t.cpu().numpy().view(B.LF_MODE_INFO_DTYPE).reshape(H // 8, W // 8)
        assert np.array_equal(lf_g["skip"], o["lf_mi"]["skip"]), "skip flags"
        if flags.apply_loop_filter:
            assert masks_equal(d.lfm_t.cpu().numpy().view(B.LF_MASK_DTYPE).reshape(o["lfm"].shape), o["lfm"]), "masks"
        rec_g = d.rec_t.cpu().numpy()
        assert np.array_equal(rec_g, o["rec"].buf), ("reconstruction", int(np.sum(rec_g != o["rec"].buf)))
        assert not inter_y.all()
