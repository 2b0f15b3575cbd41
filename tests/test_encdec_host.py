"""Host forms of the EncDec driver's device-side builders (svt-vp9_amd/csrc/encdec_core.h: ONE text compiled for the host and for
the device): transform-block lists from the mode-info grid, the stand-in decision, the stage flags, and the normative VP9 tables
the shim carries.  The GPU tests (test_gpu_encdec.py) check that the device produces exactly these lists."""
import ctypes as C
import os

import numpy as np
import pytest

import svt_testlib as T

B = T.B
_W4 = [1, 1, 2, 2, 2, 4, 4, 4, 8, 8, 8, 16, 16]
_H4 = [1, 2, 1, 2, 4, 2, 4, 8, 4, 8, 16, 8, 16]


def _uv_tx(bs, tx):
    m = min(max(_W4[bs] // 2, 1), max(_H4[bs] // 2, 1))
    return min(tx, 3 if m >= 8 else 2 if m >= 4 else 1 if m >= 2 else 0)


def _zorder(x4, y4):
    v = 0
    for b in range(4):
        v |= ((x4 >> b) & 1) << (2 * b) | ((y4 >> b) & 1) << (2 * b + 1)
    return v


def model_blocks(mi, mi_rows, mi_cols, geom, pic=0):
    """independent enumeration: list of (tx_size, plane, x, y) in the documented order"""
    W = geom.width
    sb_cols = (W + 63) // 64
    out = [[] for _ in range(4)]
    for sr in range((mi_rows + 7) // 8):
        for sc in range(sb_cols):
            for u in range(64):
                ur, uc = sr * 8 + (u >> 3), sc * 8 + (u & 7)
                if ur >= mi_rows or uc >= mi_cols:
                    continue
                bs, tx = int(mi["sb_type"][ur, uc]), int(mi["tx_size"][ur, uc])
                w8, h8 = max(_W4[bs] // 2, 1), max(_H4[bs] // 2, 1)
                if ur % h8 or uc % w8:
                    continue
                if not mi["is_inter"][ur, uc]:
                    continue                                   # intra blocks are not in the lists: the intra pass codes them
                for plane in range(3):
                    ts = _uv_tx(bs, tx) if plane else tx
                    n = 4 << ts
                    x0, y0 = (uc * 4, ur * 4) if plane else (uc * 8, ur * 8)
                    pw, ph = (w8 * 4, h8 * 4) if plane else (w8 * 8, h8 * 8)
                    for y in range(y0, y0 + ph, n):
                        for x in range(x0, x0 + pw, n):
                            out[ts].append((ts, plane, x, y, ur, uc))
    return out


def make_geom(W, H, pic=0, pad=80):
    g = B.TqPicGeom()
    pw, cpw = W + 2 * pad, W // 2 + pad
    g.width, g.height = W, H
    g.src_off[0], g.src_off[1], g.src_off[2] = 0, W * H, W * H + (W // 2) * (H // 2)
    for k in range(3):
        g.pred_off[k] = g.src_off[k] + 64
    g.src_stride[0], g.src_stride[1] = W, W // 2
    g.pred_stride[0], g.pred_stride[1] = W, W // 2
    g.recon_off[0] = pad * pw + pad
    g.recon_off[1] = pw * (H + 2 * pad) + (pad // 2) * cpw + pad // 2
    g.recon_off[2] = g.recon_off[1] + cpw * (H // 2 + pad)
    g.recon_stride[0], g.recon_stride[1] = pw, cpw
    g.coeff_base = pic * (((W + 63) // 64) * ((H + 63) // 64) * B.SB_COEFFS)
    g.recon_set, g.pic, g.do_recon = pic % 8, pic, 1
    return g


def host_blocks(mis, geoms, mi_stride, cap):
    n = len(mis)
    arr = (C.c_void_p * n)(*[m.ctypes.data for m in mis])
    gs = (B.TqPicGeom * n)(*geoms)
    blocks = np.zeros(cap, dtype=B.TQ_BLOCK_DTYPE)
    pos = np.zeros(cap, np.uint32)
    cnt = (C.c_int32 * 4)()
    rc = B.load().svt_hip_tq_blocks_from_grid(n, arr, mi_stride, gs, blocks.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p), cap, cnt)
    return rc, blocks, pos, list(cnt)


@pytest.mark.parametrize("seed,W,H", [(1, 128, 64), (2, 192, 128), (3, 64, 192)])
def test_tq_blocks_from_grid(seed, W, H):
    mi_rows, mi_cols = H // 8, W // 8
    mis, geoms, want = [], [], [[] for _ in range(4)]
    for p in range(2):
        _, _, mi = T.gen_mode_info_grid(seed * 10 + p, mi_rows, mi_cols, mi_stride=mi_cols + 3)
        mi["pad"][..., 0] = np.random.default_rng(seed + p).integers(0, 4, mi["pad"][..., 0].shape)
        mis.append(mi)
        geoms.append(make_geom(W, H, p))
        m = model_blocks(mi, mi_rows, mi_cols, geoms[-1], p)
        for s in range(4):
            want[s] += [(p,) + t for t in m[s]]
    cap = 2 * W * H * 3 // 32
    rc, blocks, pos, cnt = host_blocks(mis, geoms, mi_cols + 3, cap)
    assert rc == sum(cnt) and cnt == [len(w) for w in want]
    flat = [t for s in range(4) for t in want[s]]
    offs, _ = T.iscan_array()
    covered = [np.zeros((H * 3 // 2, W), np.int32) for _ in range(2)]
    coeff_used = np.zeros(2 * ((W + 63) // 64) * ((H + 63) // 64) * B.SB_COEFFS, np.int32)
    for i, (p, ts, plane, x, y, ur, uc) in enumerate(flat):
        k, g, n = blocks[i], geoms[p], 4 << ts
        c = 1 if plane else 0
        assert k["tx_size"] == ts and k["qtab"] == c and k["do_recon"] == 1 and k["partial32"] == 0
        assert k["src_off"] == g.src_off[plane] + y * g.src_stride[c] + x and k["pred_off"] == k["src_off"] + 64
        assert k["recon_off"] == g.recon_off[plane] + y * g.recon_stride[c] + x
        assert (k["src_stride"], k["pred_stride"], k["recon_stride"]) == (g.src_stride[c], g.pred_stride[c], g.recon_stride[c])
        tt = int(mis[p]["pad"][ur, uc, 0]) & 3 if (plane == 0 and ts < 3) else 0
        assert k["tx_type"] == tt
        assert int(pos[i]) == (tt << 30 | p << 24 | plane << 22 | (y >> 2) << 11 | (x >> 2))   # the position code IS the block (round 5)
        assert (int(k["pad"][0]) >> 4) & 7 == p and (int(k["pad"][0]) >> 3) & 1 == int(mis[p]["is_inter"][ur, uc]) and (int(k["pad"][0]) >> 2) & 1 == c
        sbw = 32 if plane else 64
        sb = (y // sbw) * ((W + 63) // 64) + x // sbw
        co = g.coeff_base + sb * B.SB_COEFFS + (0, 4096, 5120)[plane] + _zorder((x % sbw) >> 2, (y % sbw) >> 2) * 16
        assert k["coeff_off"] == co and co % 8 == 0
        coeff_used[co:co + n * n] += 1
        yy = y if plane == 0 else H + y
        xx = x if plane < 2 else W // 2 + x
        covered[p][yy:yy + n, xx:xx + n] += 1
    # every sample of the inter blocks of both pictures exactly once, nothing of the intra blocks
    for p in range(2):
        inter = np.kron(mis[p]["is_inter"][:, :mi_cols].astype(bool), np.ones((8, 8), bool))
        ci = np.kron(mis[p]["is_inter"][:, :mi_cols].astype(bool), np.ones((4, 4), bool))
        want_cov = np.zeros((H * 3 // 2, W), np.int32)
        want_cov[:H] = inter
        want_cov[H:, :W // 2] = ci
        want_cov[H:, W // 2:] = ci
        assert np.array_equal(covered[p], want_cov)
    assert (coeff_used <= 1).all() and coeff_used.sum() == sum(int(np.kron(m["is_inter"][:, :mi_cols].astype(bool), np.ones((8, 8), bool)).sum()) * 3 // 2 for m in mis)


def test_tq_blocks_iscan_offsets_match_tables():
    offs = (C.POINTER(C.c_uint32))()
    n = C.c_int32()
    tab = B.load().svt_hip_vp9_iscan_tables(C.byref(offs), C.byref(n))
    t = T.scan_tables()
    a = np.ctypeslib.as_array(tab, (n.value,))
    for ts in range(4):
        for tt in range(4):
            want = t[f"iscan_{ts}_{tt if ts < 3 else 0}"]
            o = offs[ts * 4 + tt]
            assert np.array_equal(a[o:o + want.size], want), (ts, tt)


def test_malformed_grid_is_rejected():
    W, H = 128, 64
    _, _, mi = T.gen_mode_info_grid(5, H // 8, W // 8, mi_stride=W // 8)
    mi["sb_type"][:] = 12                                       # 64x64 blocks; 64-high picture: fine
    mi["tx_size"][:] = 3
    mi["is_inter"][:] = 1
    rc, *_ = host_blocks([mi], [make_geom(W, H)], W // 8, W * H * 3 // 32)
    assert rc == 2 * (4 + 2)
    bad = mi.copy()
    bad["sb_type"][0, 0] = 3
    bad["tx_size"][0, 0] = 2                                    # 16x16 transform in an 8x8 block
    assert host_blocks([bad], [make_geom(W, H)], W // 8, W * H * 3 // 32)[0] < 0
    mi2 = np.zeros((40 // 8, W // 8), dtype=B.LF_MODE_INFO_DTYPE)
    mi2["sb_type"], mi2["tx_size"] = 12, 3                      # 64x64 blocks in a 40-row picture: cross the edge
    assert host_blocks([mi2], [make_geom(W, 40)], W // 8, W * 40 * 3 // 32)[0] < 0


def _md(results, W, H, lam=200, level=17):
    mi_rows, mi_cols = H // 8, W // 8
    mc = np.zeros((mi_rows, mi_cols), dtype=B.MC_MODE_INFO_DTYPE)
    lf = np.zeros((mi_rows, mi_cols), dtype=B.LF_MODE_INFO_DTYPE)
    rc = B.load().svt_hip_md_default_picture(results.ctypes.data_as(C.c_void_p), W, H, lam, level, mc.ctypes.data_as(C.c_void_p),
                                             lf.ctypes.data_as(C.c_void_p), mi_cols)
    assert rc == 0
    return mc, lf


@pytest.mark.parametrize("W,H", [(192, 136), (128, 64), (200, 72)])
def test_md_default_is_wellformed_and_follows_me(W, H):
    rng = np.random.default_rng(W + H)
    nsb = T.n_sb(W, H)
    res = np.zeros((nsb, 85), dtype=B.ME_RESULT_DTYPE)
    res["dist0"] = rng.integers(0, 5000, (nsb, 85))
    res["dir0"] = rng.integers(0, 3, (nsb, 85))
    for f in ("x_mv_l0", "y_mv_l0", "x_mv_l1", "y_mv_l1"):
        res[f] = rng.integers(-200, 200, (nsb, 85))
    res["dist0"][0, 0] = 0                                      # SB 0 merges to 64x64 when it fits
    res["dist0"][1 % nsb, 0:5] = 10 ** 6                        # SB 1 never merges (neither 32x32 nor 64x64)
    mc, lf = _md(res, W, H)
    rc, blocks, pos, cnt = host_blocks([lf], [make_geom(W, H)], W // 8, W * H * 3 // 32)
    assert rc > 0                                               # well-formed: every block inside the picture, aligned
    mi_rows, mi_cols = H // 8, W // 8
    if W >= 64 and H >= 64:
        assert (lf["sb_type"][:8, :8] == 12).all() and (lf["tx_size"][:8, :8] == 3).all()
    nsbx = (W + 63) // 64
    for r in range(mi_rows):
        for c in range(mi_cols):
            bs = int(lf["sb_type"][r, c])
            w8 = {12: 8, 9: 4, 6: 2, 3: 1}[bs]
            assert lf["tx_size"][r, c] == {12: 3, 9: 3, 6: 2, 3: 1}[bs] and mc["bw8"][r, c] == w8 == mc["bh8"][r, c]
            r0, c0 = r - r % w8, c - c % w8
            assert r0 + w8 <= mi_rows and c0 + w8 <= mi_cols
            q32, q16, q8 = ((r % 8) >> 2) * 2 + ((c % 8) >> 2), (((r % 8) >> 1) & 1) * 2 + (((c % 8) >> 1) & 1), (r & 1) * 2 + (c & 1)
            pu = {12: 0, 9: 1 + q32, 6: 5 + 4 * q32 + q16, 3: 21 + 16 * q32 + 4 * q16 + q8}[bs]
            rec = res[(r >> 3) * nsbx + (c >> 3), pu]
            d = int(rec["dir0"])
            assert mc["ref_list"][r, c, 0] == (1 if d == 1 else 0) and mc["ref_list"][r, c, 1] == (1 if d == 2 else -1)
            assert mc["mv_col"][r, c, 0] == 2 * (rec["x_mv_l1"] if d == 1 else rec["x_mv_l0"])
            assert mc["mv_row"][r, c, 1] == (2 * rec["y_mv_l1"] if d == 2 else 0)
            assert lf["is_inter"][r, c] == 1 and lf["filter_level"][r, c] == 17 and lf["skip"][r, c] == 0
    if nsb > 1 and W >= 128:
        assert (lf["sb_type"][:8, 8:16] != 9).all() and (lf["sb_type"][:8, 8:16] != 12).all()


def test_encdec_flags():
    lib = B.load()

    def f(**kw):
        c, o = B.EncdecFlagsConfig(**kw), B.EncdecFlags()
        assert lib.svt_hip_encdec_flags_derive(C.byref(c), C.byref(o)) == 0
        return o
    # enc-mode 8, OQ, no recon output: only base-layer pictures are deblocked; non-reference pictures are not reconstructed
    o = f(enc_mode=8, tune=1, temporal_layer_index=0, is_used_as_reference=1, recon_file=0, loop_filter=1)
    assert (o.do_recon, o.apply_loop_filter, o.pad_reference, o.allow_enc_dec_mismatch) == (1, 1, 1, 0)
    o = f(enc_mode=8, tune=1, temporal_layer_index=2, is_used_as_reference=1, recon_file=0, loop_filter=1)
    assert (o.do_recon, o.apply_loop_filter, o.pad_reference, o.allow_enc_dec_mismatch) == (1, 0, 1, 1)
    o = f(enc_mode=8, tune=1, temporal_layer_index=4, is_used_as_reference=0, recon_file=0, loop_filter=1)
    assert (o.do_recon, o.apply_loop_filter, o.pad_reference, o.limit_intra) == (0, 0, 0, 1)
    o = f(enc_mode=8, tune=1, temporal_layer_index=4, is_used_as_reference=0, recon_file=1, loop_filter=1)
    assert (o.do_recon, o.apply_loop_filter, o.pad_reference) == (1, 1, 0)
    o = f(enc_mode=3, tune=0, temporal_layer_index=3, is_used_as_reference=0, recon_file=0, loop_filter=1)
    assert (o.do_recon, o.apply_loop_filter, o.limit_intra, o.allow_enc_dec_mismatch) == (1, 0, 0, 0)
    o = f(enc_mode=5, tune=0, temporal_layer_index=3, is_used_as_reference=0, recon_file=0, loop_filter=1)
    assert o.limit_intra == 0 and f(enc_mode=5, tune=1, temporal_layer_index=3, is_used_as_reference=0, recon_file=0, loop_filter=1).limit_intra == 1
    assert f(enc_mode=9, tune=2, temporal_layer_index=2, is_used_as_reference=1, recon_file=0, loop_filter=1).apply_loop_filter == 1   # VMAF: no mismatch
    assert f(enc_mode=8, tune=1, temporal_layer_index=0, is_used_as_reference=1, recon_file=1, loop_filter=0).apply_loop_filter == 0


def test_quant_tables_for_qindex_match_init():
    lib = B.load()
    for q in (0, 40, 160, 255):
        out = (B.QuantTables * 2)()
        assert lib.svt_hip_quant_tables_for_qindex(q, out) == 0
        dc, ac = lib.svt_hip_vp9_dc_step(q), lib.svt_hip_vp9_ac_step(q)
        one = B.QuantTables()
        assert lib.svt_hip_quant_tables_init(q, dc, dc, ac, C.byref(one)) == 0
        assert bytes(out[0]) == bytes(one) == bytes(out[1])
    assert lib.svt_hip_vp9_qindex_from_qp(40) == 160 and lib.svt_hip_vp9_qindex_from_qp(63) == 255 and lib.svt_hip_vp9_qindex_from_qp(64) < 0


def _flags_vs(ref):
    lib = B.load()
    for tune in range(3):
        for mode in range(13):
            for layer in range(5):
                for used in range(2):
                    c = B.EncdecFlagsConfig(enc_mode=mode, tune=tune, temporal_layer_index=layer, is_used_as_reference=used, recon_file=0, loop_filter=1)
                    o = B.EncdecFlags()
                    assert lib.svt_hip_encdec_flags_derive(C.byref(c), C.byref(o)) == 0
                    assert (o.limit_intra, o.allow_enc_dec_mismatch) == tuple(int(v) for v in ref[tune, mode, layer, used]), (tune, mode, layer, used)


def test_encdec_flags_vs_reference_golden():
    """limit_intra / allow_enc_dec_mismatch for every (tune, enc-mode, layer, reference or not) equal what the reference's own
    eb_vp9_signal_derivation_enc_dec_kernel_{sq,oq,vmaf} derive (fixture from oracle/_ref/ref_refpad, request 'SVFL')"""
    ref = np.load(os.path.join(T.GOLDEN_DIR, "encdec_flags_reference.npz"))["flags"]
    assert ref.shape == (3, 13, 5, 2, 2) and ref[..., 0].any() and ref[..., 1].any()
    _flags_vs(ref)


@pytest.mark.skipif(not T.have_ref("ref_refpad"), reason="oracle/_ref/ref_refpad not built (reference absent)")
def test_encdec_flags_vs_reference_live():
    _flags_vs(T.ref_encdec_flags())
