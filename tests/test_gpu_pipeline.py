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
