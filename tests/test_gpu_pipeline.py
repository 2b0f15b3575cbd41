"""End-to-end chain of the stages on one picture, GPU (through the C ABI) against the oracle at every hand-over:
ME -> mode info from the ME results -> inter prediction -> transform / quantisation / reconstruction (+ distortion) ->
coefficient rate -> loop-filter masks from the mode info -> deblocking.  Checks that the data one stage writes is what
the next one reads (MV units, PU order, plane layouts, coefficient offsets, eobs, mode-info records), not only that
each stage equals its oracle in isolation."""
import ctypes as C

import numpy as np
import pytest

import me_configs as MC
import svt_testlib as T
from test_gpu_me import hip_me_picture

B = T.B
pytestmark = pytest.mark.gpu
W, H = 256, 192   # 4 x 3 superblocks


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


def _psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def test_stages_compose(ctx):
    frames = T.gen_clip_subpel(W, H, 3, 17)
    cur_y, planes = frames[1], []
    for k in (0, 2):
        u, v = _chroma(frames[k], k)
        planes.append((frames[k], u, v))
    cur_u, cur_v = _chroma(cur_y, 1)

    # ---- 1. motion estimation: both lists ----
    pics = [T.PaPic(f) for f in frames]
    p = MC.preset("c2_1080p_m8", 2, 1)
    me_o, _ = T.oracle_me_picture(pics[1], pics[0], pics[2], p)
    me_g, _ = hip_me_picture(ctx, pics[1], pics[0], pics[2], p)
    assert not T.me_results_equal(me_o, me_g, 2)

    # ---- 2. mode info: one 16x16 block per 16x16 PU, prediction direction and MVs of its best ME candidate.  PU 5 + z:
    # z = 4 * (32x32 quadrant) + (16x16 quadrant inside it), quadrants in raster order (Codec/EbMotionEstimationProcess.c:26-46);
    # ME MVs are quarter-sample, mi->mv is 1/8 sample ----
    mi_rows, mi_cols, nsbx = H // 8, W // 8, W // 64
    mi = np.zeros((mi_rows, mi_cols), dtype=B.MC_MODE_INFO_DTYPE)
    for sb in range(me_g.shape[0]):
        sx, sy = (sb % nsbx) * 8, (sb // nsbx) * 8
        for z in range(16):
            r = me_g[sb, 5 + z]
            q, k = z >> 2, z & 3
            c0, r0 = sx + 4 * (q & 1) + 2 * (k & 1), sy + 4 * (q >> 1) + 2 * (k >> 1)
            d = int(r["dir0"])          # 0 = list 0, 1 = list 1, 2 = bi-prediction
            cell = np.zeros((), dtype=B.MC_MODE_INFO_DTYPE)
            cell["bw8"] = cell["bh8"] = 2
            cell["ref_list"] = (0, -1) if d == 0 else (1, -1) if d == 1 else (0, 1)
            mvs = {0: (r["y_mv_l0"], r["x_mv_l0"]), 1: (r["y_mv_l1"], r["x_mv_l1"])}
            for j, l in enumerate([x for x in cell["ref_list"] if x >= 0]):
                cell["mv_row"][j], cell["mv_col"][j] = 2 * int(mvs[l][0]), 2 * int(mvs[l][1])
            mi[r0:r0 + 2, c0:c0 + 2] = cell
    assert len(set(me_g["dir0"][:, 5:21].ravel().tolist())) > 1    # uni- and bi-predicted blocks both occur

    # ---- 3. inter prediction from the two padded references ----
    pad = 80
    refs = [tuple(np.ascontiguousarray(np.pad(pl, pad if i == 0 else pad // 2, mode="edge")) for i, pl in enumerate(trio)) for trio in planes]
    mcase = dict(mi=mi, mi_rows=mi_rows, mi_cols=mi_cols, refs=refs, pad=pad, use_subpel=1, width=W, height=H)
    pr_o, pr_g = T.oracle_mc_frame(mcase), T.hip_mc_frame(ctx, mcase)
    for a, b in zip(pr_o, pr_g):
        assert np.array_equal(a, b)
    assert _psnr(pr_g[0], cur_y) > 28   # the ME's MVs, read correctly, predict the picture

    # ---- 4. transform / quantisation / reconstruction: 16x16 DCT on luma, 8x8 on chroma; planes in one buffer ----
    def pack(y, u, v):
        buf = np.zeros((H + H // 2, W), np.uint8)
        buf[:H] = y
        buf[H:, :W // 2] = u
        buf[H:, W // 2:] = v
        return buf
    src, pred = pack(cur_y, cur_u, cur_v), pack(*pr_g)
    iscan, offs = T.iscan_array()
    qtabs = np.array([T.quant_table(40, 48), T.quant_table(44, 52)], dtype=B.QUANT_DTYPE)
    rows = []
    for (r0, c0, hh, ww, n, ts, qi) in ((H, 0, H // 2, W // 2, 8, 1, 1), (H, W // 2, H // 2, W // 2, 8, 1, 1), (0, 0, H, W, 16, 2, 0)):
        for yy in range(0, hh, n):
            for xx in range(0, ww, n):
                rows.append((ts, (r0 + yy) * W + c0 + xx, qi))
    blocks = np.zeros(len(rows), dtype=B.TQ_BLOCK_DTYPE)
    pos = 0
    for i, (ts, off, qi) in enumerate(rows):
        n = T.TX_N[ts]
        blocks[i] = (off, off, off, pos, offs[(ts, 0)], W, W, W, ts, 0, qi, 1, 0, 0)
        pos += n * n
    counts = np.array([0, sum(1 for r in rows if r[0] == 1), sum(1 for r in rows if r[0] == 2), 0], np.int32)
    tq = dict(src=src, pred=pred, blocks=blocks, counts=counts, qtabs=qtabs, iscan=iscan, n_coeff=pos)
    tq_o, tq_g = T.oracle_tq_batch(tq), T.hip_tq_batch(ctx, tq)
    for name, a, b in zip(("recon", "qcoeff", "dqcoeff", "eob"), tq_o, tq_g):
        assert np.array_equal(a, b), name
    recon, qcoeff, _, eob = tq_g
    assert _psnr(recon[:H], cur_y) > _psnr(pred[:H], cur_y)   # the coded residual brings the picture closer to the source

    # ---- 5. coefficient rate of every block from the quantised coefficients and eobs of stage 4 ----
    roffs, _ = T.rate_scan_offsets()
    rb = np.zeros(len(blocks), dtype=B.RATE_BLOCK_DTYPE)
    rb["coeff_off"], rb["tx_size"], rb["eob"] = blocks["coeff_off"], blocks["tx_size"], eob
    rb["scan_off"] = [roffs[(int(t), 0)] for t in blocks["tx_size"]]
    rb["plane_type"] = (blocks["src_off"] >= H * W).astype(np.uint8)
    rb["is_inter"] = 1
    rcase = dict(qcoeff=qcoeff, blocks=rb)
    bits_o, bits_g = T.oracle_rate_batch(rcase), T.hip_rate_batch(ctx, rcase)
    assert np.array_equal(bits_o, bits_g) and bits_g.min() > 0 and len(set(bits_g.tolist())) > 16

    # ---- 6. loop-filter masks from the same partition (16x16 inter blocks, TX_16X16, skip = no luma coefficients) ----
    lmi = np.zeros((mi_rows, mi_cols), dtype=B.LF_MODE_INFO_DTYPE)
    lmi["sb_type"], lmi["tx_size"], lmi["is_inter"], lmi["filter_level"] = 6, 2, 1, 24
    luma_eob = eob[counts[1]:].reshape(H // 16, W // 16)
    lmi["skip"] = np.kron((luma_eob == 0).astype(np.uint8), np.ones((2, 2), np.uint8))
    lfm_o, lfm_p = T.oracle_lf_build_masks(lmi, mi_rows, mi_cols), T.product_lf_build_masks(lmi, mi_rows, mi_cols)
    assert all(np.array_equal(lfm_o[n], lfm_p[n]) for n in lfm_o.dtype.names)

    # ---- 7. deblocking of the reconstruction ----
    thr = B.LfThresh()
    B.load().svt_hip_lf_thresh_init(C.byref(thr), 0)
    lcase = dict(y=recon[:H].copy(), u=recon[H:, :W // 2].copy(), v=recon[H:, W // 2:].copy(), lfm=lfm_p, thr=thr, mi_rows=mi_rows, mi_cols=mi_cols)
    lf_o, lf_g = T.oracle_lf_frame(lcase), T.hip_lf_frame(ctx, lcase)
    for a, b in zip(lf_o, lf_g):
        assert np.array_equal(a, b)
    assert (lf_g[0] != lcase["y"]).any() and _psnr(lf_g[0], cur_y) > 30


# ---------------------------------------------------------------------------------------------------------------------
# the closed loop: picture B is predicted from the DEBLOCKED, PADDED RECONSTRUCTION of picture A, everything device resident
# ---------------------------------------------------------------------------------------------------------------------
PAD = 80


def _mi_from_me(me, nl):
    """one 16x16 block per 16x16 PU: direction and MVs of its best ME candidate (as in test_stages_compose)"""
    mi_rows, mi_cols, nsbx = H // 8, W // 8, W // 64
    mi = np.zeros((mi_rows, mi_cols), dtype=B.MC_MODE_INFO_DTYPE)
    for sb in range(me.shape[0]):
        sx, sy = (sb % nsbx) * 8, (sb // nsbx) * 8
        for z in range(16):
            r = me[sb, 5 + z]
            q, k = z >> 2, z & 3
            c0, r0 = sx + 4 * (q & 1) + 2 * (k & 1), sy + 4 * (q >> 1) + 2 * (k >> 1)
            d = int(r["dir0"]) if nl == 2 else 0
            cell = np.zeros((), dtype=B.MC_MODE_INFO_DTYPE)
            cell["bw8"] = cell["bh8"] = 2
            cell["ref_list"] = (0, -1) if d == 0 else (1, -1) if d == 1 else (0, 1)
            mvs = {0: (r["y_mv_l0"], r["x_mv_l0"]), 1: (r["y_mv_l1"], r["x_mv_l1"])}
            for j, l in enumerate([x for x in cell["ref_list"] if x >= 0]):
                cell["mv_row"][j], cell["mv_col"][j] = 2 * int(mvs[l][0]), 2 * int(mvs[l][1])
            mi[r0:r0 + 2, c0:c0 + 2] = cell
    return mi


class _RefLayout:
    """a padded reference picture as three planes in one buffer (the layout bench.py uses)"""
    pw, ph, cpw, cph = W + 2 * PAD, H + 2 * PAD, W // 2 + PAD, H // 2 + PAD
    u_base = pw * ph
    v_base = u_base + cpw * cph
    size = v_base + cpw * cph
    y0, u0, v0 = PAD * pw + PAD, u_base + (PAD // 2) * cpw + PAD // 2, v_base + (PAD // 2) * cpw + PAD // 2

    @classmethod
    def planes(cls, buf):
        return (buf[:cls.u_base].reshape(cls.ph, cls.pw), buf[cls.u_base:cls.v_base].reshape(cls.cph, cls.cpw),
                buf[cls.v_base:cls.size].reshape(cls.cph, cls.cpw))

    @classmethod
    def from_planes(cls, y, u, v, pad_edges):
        buf = np.zeros(cls.size, np.uint8)
        py, pu, pv = cls.planes(buf)
        for dst, src, pd in ((py, y, PAD), (pu, u, PAD // 2), (pv, v, PAD // 2)):
            if pad_edges:
                dst[:] = np.pad(src, pd, mode="edge")
            else:
                dst[pd:pd + src.shape[0], pd:pd + src.shape[1]] = src
        return buf


def test_closed_loop_two_temporal_levels(ctx):
    """A (lower temporal layer) is predicted from the previous base picture, coded, deblocked and padded IN its reference buffer;
    B is then predicted from that buffer (list 1) and the previous base (list 0), coded, deblocked and padded.  GPU: every
    stage through the *_device entry points on buffers that never leave the device (the transform stage reconstructs straight into
    the padded reference buffer, the loop filter and svt_hip_ref_pad_batch_device work in place, the prediction of B reads what
    they left) -- against the oracle's chain stage by stage.  This is the data dependency bench.py's waves are built on:
    svt_lf_kernel -> svt_refpad_kernel -> svt_mc_kernel of the next temporal layer (Codec/EbEncDecProcess.c:5676-5696 ->
    4822-4851 -> the next picture's inter_prediction)."""
    import torch
    lib = B.load()
    dev = torch.device("cuda", 0)
    L = _RefLayout
    frames = T.gen_clip_subpel(W, H, 3, 29)
    srcs = [frames[2], frames[1]]                       # coding order: A = frame 2 (from frame 0), B = frame 1 (from frame 0 and A)
    chroma = [_chroma(f, k) for k, f in enumerate(frames)]
    pics = [T.PaPic(f) for f in frames]
    mi_rows, mi_cols = H // 8, W // 8
    # motion estimation on the source pictures (the ME side of the path never sees a reconstruction)
    me_a, _ = hip_me_picture(ctx, pics[2], pics[0], None, MC.preset("c2_1080p_m8", 1, 0))
    me_b, _ = hip_me_picture(ctx, pics[1], pics[0], pics[2], MC.preset("c2_1080p_m8", 2, 1))
    mis = [_mi_from_me(me_a, 1), _mi_from_me(me_b, 2)]
    assert (mis[1]["ref_list"][..., 1] >= 0).any() or (mis[1]["ref_list"][..., 0] == 1).any()     # B really uses A

    def pack(y, u, v):
        buf = np.zeros((H + H // 2, W), np.uint8)
        buf[:H], buf[H:, :W // 2], buf[H:, W // 2:] = y, u, v
        return buf
    src_t = [pack(srcs[k], *chroma[2 - k]) for k in range(2)]
    iscan, offs = T.iscan_array()
    qtabs = np.array([T.quant_table(40, 48), T.quant_table(44, 52)], dtype=B.QUANT_DTYPE)
    rows = []
    for (r0, c0, hh, ww, n, ts, qi) in ((H, 0, H // 2, W // 2, 8, 1, 1), (H, W // 2, H // 2, W // 2, 8, 1, 1), (0, 0, H, W, 16, 2, 0)):
        for yy in range(0, hh, n):
            for xx in range(0, ww, n):
                rows.append((ts, (r0 + yy) * W + c0 + xx, qi, r0 + yy, c0 + xx))
    nblk = len(rows)
    tight = np.zeros(nblk, dtype=B.TQ_BLOCK_DTYPE)
    pos = 0
    for i, (ts, off, qi, _, _) in enumerate(rows):
        tight[i] = (off, off, off, pos, offs[(ts, 0)], W, W, W, ts, 0, qi, 1, 0, 0)
        pos += T.TX_N[ts] ** 2
    n_coeff = pos
    counts = np.array([0, sum(1 for r in rows if r[0] == 1), sum(1 for r in rows if r[0] == 2), 0], np.int32)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)

    def lf_masks(eob):
        lmi = np.zeros((mi_rows, mi_cols), dtype=B.LF_MODE_INFO_DTYPE)
        lmi["sb_type"], lmi["tx_size"], lmi["is_inter"], lmi["filter_level"] = 6, 2, 1, 24
        lmi["skip"] = np.kron((eob[counts[1]:].reshape(H // 16, W // 16) == 0).astype(np.uint8), np.ones((2, 2), np.uint8))
        return T.product_lf_build_masks(lmi, mi_rows, mi_cols)

    # ------------------------------------------------ oracle chain ------------------------------------------------
    ref0 = L.from_planes(frames[0], *chroma[0], pad_edges=True)
    o_rec, o_pred = [ref0], []
    for k in range(2):
        refs = [L.planes(o_rec[0]), L.planes(o_rec[1] if k == 1 else o_rec[0])]
        mcase = dict(mi=mis[k], mi_rows=mi_rows, mi_cols=mi_cols, refs=[tuple(np.ascontiguousarray(p) for p in r) for r in refs], pad=PAD, use_subpel=1,
                     width=W, height=H)
        pred = pack(*T.oracle_mc_frame(mcase))
        recon, _, _, eob = T.oracle_tq_batch(dict(src=src_t[k], pred=pred, blocks=tight, counts=counts, qtabs=qtabs, iscan=iscan, n_coeff=n_coeff))
        lfm = lf_masks(eob)
        y, u, v = T.oracle_lf_frame(dict(y=recon[:H].copy(), u=recon[H:, :W // 2].copy(), v=recon[H:, W // 2:].copy(), lfm=lfm, thr=thr, mi_rows=mi_rows,
                                         mi_cols=mi_cols))
        buf = L.from_planes(y, u, v, pad_edges=False)
        buf[:] = np.concatenate([p.ravel() for p in T.oracle_ref_pad(dict(bufs=[p.copy() for p in L.planes(buf)], width=W, height=H, pad_x=PAD, pad_y=PAD, slack=0))])
        o_rec.append(buf)
        o_pred.append(pred)

    # ------------------------------------------------- GPU chain --------------------------------------------------
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
    d_src, d_pred = up(np.stack(src_t)), torch.zeros(2 * src_t[0].size, dtype=torch.uint8, device=dev)
    d_rec = torch.full((3 * L.size,), 0x33, dtype=torch.uint8, device=dev)      # junk in the borders until the padding writes them
    d_rec[:L.size] = up(ref0)
    d_qt, d_is = up(qtabs), up(iscan)
    d_q, d_dq = torch.zeros(2 * n_coeff, dtype=torch.int16, device=dev), torch.zeros(2 * n_coeff, dtype=torch.int16, device=dev)
    cnt = (C.c_int32 * 4)(*[int(c_) for c_ in counts])
    p_ = lambda t: C.c_void_p(t.data_ptr())
    pic_bytes = src_t[0].size
    g_pred = []

    def yuv(slot):
        d = B.YuvPlanes()
        base = d_rec.data_ptr() + slot * L.size
        d.y, d.u, d.v, d.y_stride, d.uv_stride, d.width, d.height = base + L.y0, base + L.u0, base + L.v0, L.pw, L.cpw, W, H
        return d
    for k in range(2):
        blocks = tight.copy()
        blocks["src_off"] += np.uint32(k * pic_bytes)
        blocks["pred_off"] += np.uint32(k * pic_bytes)
        blocks["coeff_off"] += np.uint32(k * n_coeff)
        for i, (ts, off, qi, r, c) in enumerate(rows):       # reconstruct straight into the padded reference buffer of slot k + 1
            if r < H:
                blocks["recon_off"][i], blocks["recon_stride"][i] = (k + 1) * L.size + L.y0 + r * L.pw + c, L.pw
            else:
                isv = c >= W // 2
                blocks["recon_off"][i] = (k + 1) * L.size + (L.v0 if isv else L.u0) + (r - H) * L.cpw + (c - W // 2 if isv else c)
                blocks["recon_stride"][i] = L.cpw
        d_blocks, d_mi = up(blocks), up(mis[k])
        mp = (B.McPicture * 1)()
        mp[0].d_mi, mp[0].mi_stride, mp[0].mi_rows, mp[0].mi_cols, mp[0].use_subpel = d_mi.data_ptr(), mi_cols, mi_rows, mi_cols, 1
        mp[0].ref[0], mp[0].ref[1] = yuv(0), yuv(1 if k == 1 else 0)
        pb = d_pred.data_ptr() + k * pic_bytes
        mp[0].pred.y, mp[0].pred.u, mp[0].pred.v = pb, pb + H * W, pb + H * W + W // 2
        mp[0].pred.y_stride, mp[0].pred.uv_stride, mp[0].pred.width, mp[0].pred.height = W, W, W, H
        d_eob = torch.zeros(nblk, dtype=torch.int16, device=dev)
        B.check(lib.svt_hip_inter_pred_batch_device(ctx, 1, mp))
        B.check(lib.svt_hip_tq_batch_dist_device(ctx, p_(d_src), p_(d_pred), p_(d_rec), p_(d_blocks), cnt, p_(d_qt), p_(d_is), p_(d_q), p_(d_dq), p_(d_eob), None))
        B.check(lib.svt_hip_ctx_synchronize(ctx))
        lfm = lf_masks(d_eob.cpu().numpy().view(np.uint16))     # skip flags: mode decision's output, host side
        d_lfm = up(lfm)
        ydesc = (B.YuvPlanes * 1)(yuv(k + 1))
        B.check(lib.svt_hip_lf_batch_device(ctx, 1, ydesc, (C.c_void_p * 1)(d_lfm.data_ptr()), (C.c_int32 * 1)((mi_cols + 7) // 8), C.byref(thr),
                                            (C.c_int32 * 1)(mi_rows), (C.c_int32 * 1)(mi_cols), 0))
        B.check(lib.svt_hip_ref_pad_batch_device(ctx, 1, ydesc, PAD, PAD))
        B.check(lib.svt_hip_ctx_synchronize(ctx))
        g_pred.append(d_pred[k * pic_bytes:(k + 1) * pic_bytes].cpu().numpy().reshape(H + H // 2, W))
    g_rec = d_rec.cpu().numpy()
    for k in range(2):
        assert np.array_equal(g_pred[k], o_pred[k]), f"prediction of picture {k}"
        assert np.array_equal(g_rec[(k + 1) * L.size:(k + 2) * L.size], o_rec[k + 1]), f"padded reference picture {k}"
    # the loop matters: predicting B from A's SOURCE instead of A's deblocked reconstruction gives another prediction
    src_a = L.from_planes(frames[2], *chroma[2], pad_edges=True)
    refs = [L.planes(ref0), L.planes(src_a)]
    alt = pack(*T.oracle_mc_frame(dict(mi=mis[1], mi_rows=mi_rows, mi_cols=mi_cols, refs=[tuple(np.ascontiguousarray(p) for p in r) for r in refs], pad=PAD,
                                      use_subpel=1, width=W, height=H)))
    assert (alt != o_pred[1]).any()
    assert _psnr(L.planes(o_rec[2])[0][PAD:PAD + H, PAD:PAD + W], frames[1]) > 30
