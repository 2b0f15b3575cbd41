"""The stages behind mode decision through the PUBLIC API (libSvtVp9Enc.so): a clip goes in through eb_vp9_svt_enc_send_picture,
eb_vp9_svt_get_recon hands out the reconstructed pictures with the reference's semantics (coding order, pts = picture number, EOS on
the last; Codec/EbEncHandle.c:2837-2865), and every byte equals the oracle chain run on the host in the same dependency order:
each picture predicted from the RECONSTRUCTED, deblocked (where the reference deblocks), padded pictures before it.  Both decision
sources: the built-in stand-in and a host callback (svt_vp9_shim_set_mode_decision)."""
import ctypes as C
import os

import numpy as np
import pytest

import encdec_model as M
import svt_testlib as T
from test_enc_shim import Cfg, PicInfo, shim

B = T.B
pytestmark = pytest.mark.gpu


class In(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("luma", "cb", "cr", "luma_ext", "cb_ext", "cr_ext")] + [(n, C.c_uint32) for n in ("y_stride", "cr_stride", "cb_stride")]


class Hdr(C.Structure):
    _fields_ = [("size", C.c_uint32), ("p_buffer", C.c_void_p), ("n_filled_len", C.c_uint32), ("n_alloc_len", C.c_uint32), ("p_app_private", C.c_void_p),
                ("wrapper_ptr", C.c_void_p), ("n_tick_count", C.c_uint32), ("dts", C.c_int64), ("pts", C.c_int64), ("qp", C.c_uint32), ("pic_type", C.c_uint32),
                ("flags", C.c_uint32)]


MD_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(PicInfo), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32)
EMPTY = 0x80002033


def chroma(y, n):
    return (y[::2, ::2] // 2 + 32 + (n % 5)).astype(np.uint8), (200 - y[1::2, ::2] // 3).astype(np.uint8)


def host_decision(me, W, H, number, level):
    """a host-side 'mode decision' for the callback test: a seeded partition per 32x32 area among {8x8 blocks / 4x4 transforms, 8x8 /
    8x8, 16x16 / 16x16, 32x32 / 32x32}, every block with the best ME candidate of its own PU (the rule bench.py uses)"""
    mi_rows, mi_cols, nsbx = H // 8, W // 8, (W + 63) // 64
    kinds = np.random.default_rng(1000 + number).integers(0, 4, ((H + 31) // 32, (W + 31) // 32))
    r, c = np.meshgrid(np.arange(mi_rows), np.arange(mi_cols), indexing="ij")
    k = kinds[r >> 2, c >> 2]
    q32, q16, q8 = (((r >> 2) & 1) * 2 + ((c >> 2) & 1)), (((r >> 1) & 1) * 2 + ((c >> 1) & 1)), ((r & 1) * 2 + (c & 1))
    pu = np.where(k == 3, 1 + q32, np.where(k == 2, 5 + 4 * q32 + q16, 21 + 16 * q32 + 4 * q16 + q8))
    rec = me[(r >> 3) * nsbx + (c >> 3), pu]
    d = rec["dir0"].astype(np.int64)
    mc = np.zeros((mi_rows, mi_cols), dtype=B.MC_MODE_INFO_DTYPE)
    bw = np.where(k == 3, 4, np.where(k == 2, 2, 1)).astype(np.uint8)
    mc["bw8"], mc["bh8"] = bw, bw
    mc["ref_list"][..., 0] = np.where(d == 1, 1, 0)
    mc["ref_list"][..., 1] = np.where(d == 2, 1, -1)
    mc["mv_row"][..., 0] = 2 * np.where(d == 1, rec["y_mv_l1"], rec["y_mv_l0"])
    mc["mv_col"][..., 0] = 2 * np.where(d == 1, rec["x_mv_l1"], rec["x_mv_l0"])
    mc["mv_row"][..., 1] = np.where(d == 2, 2 * rec["y_mv_l1"].astype(np.int32), 0)
    mc["mv_col"][..., 1] = np.where(d == 2, 2 * rec["x_mv_l1"].astype(np.int32), 0)
    lf = np.zeros((mi_rows, mi_cols), dtype=B.LF_MODE_INFO_DTYPE)
    lf["sb_type"] = np.where(k == 3, 9, np.where(k == 2, 6, 3))
    lf["tx_size"], lf["is_inter"], lf["filter_level"] = k, 1, level
    if number % 5 == 2:                                        # some pictures carry intra blocks (random modes) among the inter ones
        lf, mc, _ = M.make_mixed(3000 + number, lf, mc, share=0.2, level=level)
    return mc, lf


def intra_stand_in(W, H, level):
    """the library's stand-in for an intra picture: 16x16 blocks with DC prediction, 8x8 where a 16x16 block would cross the edge"""
    mi_rows, mi_cols = H // 8, W // 8
    r, c = np.meshgrid(np.arange(mi_rows), np.arange(mi_cols), indexing="ij")
    fit = ((r & ~1) + 2 <= mi_rows) & ((c & ~1) + 2 <= mi_cols)
    lf = np.zeros((mi_rows, mi_cols), dtype=B.LF_MODE_INFO_DTYPE)
    lf["sb_type"], lf["tx_size"], lf["filter_level"] = np.where(fit, 6, 3), np.where(fit, 2, 1), level
    return lf


def intra_host_decision(W, H, number, level):
    return M.gen_intra_grid(2000 + number, W, H, filter_level=level)


def stand_in(me, W, H, lam, level):
    mc = np.zeros((H // 8, W // 8), dtype=B.MC_MODE_INFO_DTYPE)
    lf = np.zeros((H // 8, W // 8), dtype=B.LF_MODE_INFO_DTYPE)
    assert B.load().svt_hip_md_default_picture(me.ctypes.data_as(C.c_void_p), W, H, lam, level, mc.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p), W // 8) == 0
    return mc, lf


def run_clip(W, H, N, enc_mode, tune, qp, recon_file, intra_period, use_callback, seed=41, env=None, frames=None):
    """returns (frames, delivered reconstructions {pts: bytes}, order of delivery, per-picture info, padded reference pictures of some pictures)"""
    lib = shim()
    for f_ in ("svt_vp9_shim_get_me_results", "svt_vp9_shim_get_coded_picture", "svt_vp9_shim_get_reference_picture", "svt_vp9_shim_set_mode_decision", "eb_vp9_svt_get_packet",
               "eb_vp9_svt_enc_send_picture", "eb_vp9_svt_get_recon"):
        getattr(lib, f_).restype = C.c_int32
    lib.eb_vp9_svt_release_out_buffer.restype = None
    frames = T.gen_clip_subpel(W, H, N, seed) if frames is None else frames
    cfg, h = Cfg(), C.c_void_p()
    assert lib.eb_vp9_svt_init_handle(C.byref(h), None, C.byref(cfg)) == 0
    cfg.source_width, cfg.source_height, cfg.enc_mode, cfg.tune, cfg.frame_rate, cfg.intra_period, cfg.qp, cfg.recon_file = W, H, enc_mode, tune, 60 << 16, intra_period, qp, recon_file
    assert lib.eb_vp9_svt_enc_set_parameter(h, C.byref(cfg)) == 0
    level = B.load().svt_hip_lf_level_from_q(B.load().svt_hip_vp9_ac_step(B.load().svt_hip_vp9_qindex_from_qp(qp)), 0)
    level_key = B.load().svt_hip_lf_level_from_q(B.load().svt_hip_vp9_ac_step(B.load().svt_hip_vp9_qindex_from_qp(qp)), 1)
    nsb = T.n_sb(W, H)
    seen = []

    def cb(user, info, me_p, mc_p, lf_p, mi_stride):
        i = info.contents
        seen.append(int(i.picture_number))
        if i.picture_number % 7 == 3:
            return 1                                            # "no decision": the library's stand-in takes this picture
        if i.is_intra:                                          # an intra picture: blocks and modes, no ME results
            assert not me_p
            lf = intra_host_decision(W, H, int(i.picture_number), level_key)
            C.memmove(lf_p, lf.ctypes.data, lf.nbytes)
            return 0
        me = np.ctypeslib.as_array(C.cast(me_p, C.POINTER(C.c_uint8)), (nsb * 85 * 40,)).view(B.ME_RESULT_DTYPE).reshape(nsb, 85)
        mc, lf = host_decision(me, W, H, int(i.picture_number), int(i.filter_level))   # the picture's own level: its q index is scaled by its layer
        C.memmove(mc_p, mc.ctypes.data, mc.nbytes)
        C.memmove(lf_p, lf.ctypes.data, lf.nbytes)
        return 0
    keep = MD_CB(cb)
    if use_callback:
        assert lib.svt_vp9_shim_set_mode_decision(h, keep, None) == 0
    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        assert lib.eb_vp9_init_encoder(h) == 0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    recon, order, flags_seen, packets = {}, [], [], []
    rbuf = np.zeros(W * H * 3 // 2, np.uint8)

    def poll_recon():
        while True:
            b = Hdr(size=C.sizeof(Hdr), p_buffer=rbuf.ctypes.data, n_alloc_len=rbuf.size)
            rc = lib.eb_vp9_svt_get_recon(h, C.byref(b)) & 0xffffffff
            if rc == EMPTY:
                return
            assert rc == 0, hex(rc)
            assert b.n_filled_len == rbuf.size
            recon[int(b.pts)] = rbuf.copy()
            order.append(int(b.pts))
            flags_seen.append(int(b.flags))

    def poll_packets(done):
        while True:
            pp = C.POINTER(Hdr)()
            rc = lib.eb_vp9_svt_get_packet(h, C.byref(pp), C.c_uint8(done)) & 0xffffffff
            if rc == EMPTY:
                return
            assert rc == 0
            packets.append((int(pp.contents.pts), int(pp.contents.flags)))
            lib.eb_vp9_svt_release_out_buffer(C.byref(pp))
    infos, refpics = {}, {}
    for n in range(N):
        y = np.ascontiguousarray(frames[n])
        u, v = (np.ascontiguousarray(p) for p in chroma(y, n))
        i = In(y.ctypes.data, u.ctypes.data, v.ctypes.data, None, None, None, W, W // 2, W // 2)
        b = Hdr(size=C.sizeof(Hdr), p_buffer=C.addressof(i), pts=n, flags=1 if n == N - 1 else 0)
        assert lib.eb_vp9_svt_enc_send_picture(h, C.byref(b)) == 0
        poll_packets(0)
        if recon_file:
            poll_recon()
    poll_packets(1)
    if recon_file:
        for _ in range(2000):
            poll_recon()
            if len(recon) == N:
                break
    else:
        b = Hdr(size=C.sizeof(Hdr), p_buffer=rbuf.ctypes.data, n_alloc_len=rbuf.size)
        assert lib.eb_vp9_svt_get_recon(h, C.byref(b)) == 0x7FFFFFFF        # recon is not enabled: EB_ErrorMax, as the reference
    rp = M.RefPic(W, H)
    for k in range(max(0, N - (16 if tune == 0 else 20)), N):      # (the library keeps 2 mini-GOPs + 2 pictures)
        info = PicInfo()
        mc = np.zeros((H // 8, W // 8), dtype=B.MC_MODE_INFO_DTYPE)
        lf = np.zeros((H // 8, W // 8), dtype=B.LF_MODE_INFO_DTYPE)
        q = np.zeros(nsb * B.SB_COEFFS, np.int16)
        em = np.zeros(M.eob_map_offsets(W, H)[3], np.uint16)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        assert lib.svt_vp9_shim_get_coded_picture(h, C.c_uint64(k), C.byref(info), vp(mc), vp(lf), vp(q), vp(em)) == 0
        infos[k] = dict(info=info, mc=mc, lf=lf, q=q, em=em)
        out = np.zeros(rp.u_base + 2 * rp.cpw * rp.cph, np.uint8)
        assert lib.svt_vp9_shim_get_reference_picture(h, C.c_uint64(k), vp(out), C.c_uint64(out.size)) == 0
        refpics[k] = out
    assert lib.eb_vp9_deinit_encoder(h) == 0 and lib.eb_vp9_deinit_handle(h) == 0
    return frames, recon, order, flags_seen, packets, infos, refpics, seen


def structure(N, minigop, intra_period):
    """coding structure the library documents: (number, layer, levels, n_lists, ref0, ref1, used_as_ref) in dependency order"""
    out = []
    levels = {16: 4, 8: 3}[minigop]

    def hierarchy(lo, hi, layer, lv):
        if hi - lo < 2:
            return
        mid = (lo + hi) // 2
        out.append((mid, layer, lv, 2, lo, hi, int(layer < lv)))
        hierarchy(lo, mid, layer + 1, lv)
        hierarchy(mid, hi, layer + 1, lv)

    def group(first, count, prev, cut_by_intra):
        parts = T.product_minigop_split(count + (1 if cut_by_intra else 0), levels, 1 if cut_by_intra else 0)
        for idx, (start, length, lv, ra) in enumerate(parts):
            if cut_by_intra and idx == len(parts) - 1:
                length -= 1
            if length < 1:
                continue
            p0, base = first + start, first + start + length - 1
            if ra and prev >= 0:
                out.append((base, 0, lv, 2, prev, prev, 1))
                hierarchy(prev, base, 1, lv)
            else:
                for q in range(p0, base + 1):
                    out.append((q, 0, lv, 1, q - 1, -1, 1))
            prev = base
        return prev
    n, prev, pend_first, pend = 0, -1, 0, 0
    while n < N:
        intra = n == 0 or (intra_period >= 0 and n % (intra_period + 1) == 0)
        if intra:
            if pend:
                group(pend_first, pend, prev, True)
                pend = 0
            out.append((n, 0, levels, 0, -1, -1, 1))
            prev = n
        else:
            if not pend:
                pend_first = n
            pend += 1
            if pend == minigop:
                prev = group(pend_first, pend, prev, False)
                pend = 0
        n += 1
    if pend:
        group(pend_first, pend, prev, False)
    return out


def oracle_clip(frames, W, H, N, enc_mode, tune, qp, recon_file, intra_period, use_callback):
    lib = B.load()
    q_index = lib.svt_hip_vp9_qindex_from_qp(qp)
    ac = lib.svt_hip_vp9_ac_step(q_index)
    level, level_key = lib.svt_hip_lf_level_from_q(ac, 0), lib.svt_hip_lf_level_from_q(ac, 1)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    pics = [T.PaPic(f) for f in frames]
    minigop = 16 if tune != 0 else 8
    recs, outs = {}, {}
    for (k, layer, lv, nl, r0, r1, used) in structure(N, minigop, intra_period):
        src = (frames[k],) + chroma(frames[k], k)
        if nl == 0:
            lf = intra_host_decision(W, H, k, level_key) if use_callback and k % 7 != 3 else intra_stand_in(W, H, level_key)
            c, fl = B.EncdecFlagsConfig(enc_mode=enc_mode, tune=tune, temporal_layer_index=0, is_used_as_reference=1, recon_file=recon_file, loop_filter=1), B.EncdecFlags()
            assert lib.svt_hip_encdec_flags_derive(C.byref(c), C.byref(fl)) == 0
            o = M.oracle_intra_chain(src, lf, q_index, fl, thr)
            recs[k] = o["rec"]
            outs[k] = dict(intra=True, o=o, flags=fl, layer=0)
            continue
        p = B.me_params_derive(pic_width=W, pic_height=H, enc_mode=enc_mode, tune=tune, frame_rate=60, num_ref_lists=nl, temporal_layer_index=layer,
                               hierarchical_levels=lv, is_used_as_reference=used, same_ref_poc=int(nl == 2 and r0 == r1))
        me, _ = T.oracle_me_picture_mt(pics[k], pics[r0], pics[r1] if nl == 2 else None, p)
        # fixed-QP mode: the picture's temporal layer scales the sequence QP (svt_hip_vp9_layer_qindex, pinned by tests/test_qp_scaling.py)
        q_pic = lib.svt_hip_vp9_layer_qindex(qp, tune, lv, layer, 0)
        ac_pic = lib.svt_hip_vp9_ac_step(q_pic)
        level_pic = lib.svt_hip_lf_level_from_q(ac_pic, 0)
        if use_callback and k % 7 != 3:
            mc, lf = host_decision(me, W, H, k, level_pic)
        else:
            mc, lf = stand_in(me, W, H, 4 * ac_pic, level_pic)
        c, fl = B.EncdecFlagsConfig(enc_mode=enc_mode, tune=tune, temporal_layer_index=layer, is_used_as_reference=used, recon_file=recon_file, loop_filter=1), B.EncdecFlags()
        assert lib.svt_hip_encdec_flags_derive(C.byref(c), C.byref(fl)) == 0
        o = M.oracle_encdec_picture(src, [recs[r0], recs[r1 if nl == 2 else r0]], mc, lf, q_pic, fl, thr, use_subpel=int(p.fractional_search_model != 2))
        recs[k] = o["rec"]
        outs[k] = dict(intra=False, o=o, mc=mc, flags=fl, layer=layer, refs=(r0, r1), nl=nl, q_index=q_pic, filter_level=level_pic)
    return recs, outs


@pytest.mark.parametrize("use_callback,N,intra_period", [(False, 36, -1), (True, 36, -1), (False, 34, 19), (True, 34, 19)])
def test_get_recon_equals_the_oracle_chain(use_callback, N, intra_period):
    W, H, enc_mode, tune, qp = 256, 192, 8, 1, 40
    frames, recon, order, flags_seen, packets, infos, refpics, seen = run_clip(W, H, N, enc_mode, tune, qp, 1, intra_period, use_callback)
    recs, outs = oracle_clip(frames, W, H, N, enc_mode, tune, qp, 1, intra_period, use_callback)
    assert sorted(order) == list(range(N)) and len(packets) == N and packets[-1][1] & 1          # every picture once; EOS packet last
    assert flags_seen[-1] == 1 and all(f == 0 for f in flags_seen[:-1])                            # EOS on the last reconstruction delivered
    # coding order, as recon_output posts them: the reference posts a picture when its EncDec finishes, so any order in which every
    # picture follows its reference pictures is the reference's
    at = {k: i for i, k in enumerate(order)}
    for (k, layer, lv, nl, r0, r1, used) in structure(N, 16, intra_period):
        for r in ((r0, r1) if nl == 2 else (r0,) if nl == 1 else ()):
            assert at[r] < at[k], (k, r)
    for k in range(N):
        y, u, v = recs[k].interior()
        want = np.concatenate([y.ravel(), u.ravel(), v.ravel()])
        assert np.array_equal(recon[k], want), (k, int(np.sum(recon[k] != want)), outs[k].get("layer"))
    for k, got in refpics.items():                                                                   # the padded reference pictures later pictures read
        if infos[k]["info"].pad_reference:
            assert np.array_equal(got, recs[k].buf[:got.size]), k
    n_cb = 0
    for k, d in infos.items():
        i = d["info"]
        o = outs[k]["o"]
        if outs[k]["intra"]:
            assert i.is_intra and not i.intra_recon_is_source and i.decision_source == (1 if use_callback and k % 7 != 3 else 0)
            assert (i.do_recon, i.apply_loop_filter, i.pad_reference) == (1, 1, 1)
            assert d["lf"].tobytes() == o["lf_mi"].tobytes() and np.array_equal(d["q"], o["qcoeff"]) and np.array_equal(d["em"], o["eob_map"]), k
            assert not d["mc"].view(np.uint8).any()
            continue
        assert (i.do_recon, i.apply_loop_filter, i.pad_reference) == (outs[k]["flags"].do_recon, outs[k]["flags"].apply_loop_filter, outs[k]["flags"].pad_reference)
        assert i.decision_source == (1 if use_callback and k % 7 != 3 else 0)
        n_cb += i.decision_source == 1
        assert d["mc"].tobytes() == outs[k]["mc"].tobytes() and d["lf"].tobytes() == o["lf_mi"].tobytes(), k      # incl. the skip flags
        assert np.array_equal(d["q"], o["qcoeff"]) and np.array_equal(d["em"], o["eob_map"]), k
    if use_callback:
        assert n_cb >= 10 and sorted(seen) == list(range(N))                                        # once per picture, intra pictures included
    # the loop is closed: a picture predicted from reconstructions differs from one predicted from sources
    deblocked = [k for k in range(N) if outs[k]["flags"].apply_loop_filter]
    assert len(deblocked) == N                                                                       # recon output on: every picture is deblocked


@pytest.mark.parametrize("env,N,intra_period", [
    ({"SVT_HIP_DEVICES": "0,0"}, 34, 19),                                  # two closed GOPs on two contexts (GOP g -> device g mod N)
    ({"SVT_HIP_DEVICES": "0,0", "SVT_HIP_SPLIT_GOP": "1"}, 36, -1),        # one GOP, consecutive mini-GOPs on alternating contexts + hand-off
    ({"SVT_HIP_DEVICES": "0,0,0", "SVT_HIP_SPLIT_GOP": "1"}, 34, 19),
    # short GOPs dealt to two contexts with the smallest picture ring: every device's feeder thread runs beside the caller while its slots
    # are reused (cut groups, intra refreshes enqueued by the caller between the feeder's groups)
    ({"SVT_HIP_DEVICES": "0,0", "SVT_HIP_RING_GROUPS": "2"}, 90, 9),
    ({"SVT_HIP_DEVICES": "0,0,0", "SVT_HIP_RING_GROUPS": "2"}, 75, 24),
])
def test_several_contexts_give_the_same_reconstruction(env, N, intra_period):
    """the multi-device paths of the library on one GPU: every "device" is a context of its own on ordinal 0 (own stream, own picture
    ring, own workspace); GOP sharding needs no exchange, split-GOP mode hands the padded base-layer reconstruction and its analysed
    planes from context to context (svt_hip_ref_handoff_device).  The reconstruction must not depend on how the work was dealt."""
    W, H, enc_mode, tune, qp = (256, 192, 8, 1, 40) if N <= 40 else (136, 72, 8, 1, 40)
    clip = None
    if N > 40:   # (the moving-texture clip holds 30 pictures: forth and back)
        base = T.gen_clip_subpel(W, H, 30, 57)
        clip = [base[i % 30] if (i // 30) % 2 == 0 else base[29 - i % 30] for i in range(N)]
    frames, recon, order, flags_seen, packets, infos, refpics, _ = run_clip(W, H, N, enc_mode, tune, qp, 1, intra_period, False, env=env, frames=clip)
    recs, outs = oracle_clip(frames, W, H, N, enc_mode, tune, qp, 1, intra_period, False)
    assert sorted(order) == list(range(N)) and len(packets) == N and packets[-1][1] & 1 and flags_seen[-1] == 1
    for k in range(N):
        y, u, v = recs[k].interior()
        assert np.array_equal(recon[k], np.concatenate([y.ravel(), u.ravel(), v.ravel()])), k


def test_warm_up_encoder_leaves_the_stream_alone():
    """eb_vp9_init_encoder runs a throw-away encoder on a small picture so that the first send_picture does not pay for code loading
    (SVT_HIP_WARMUP=0: off).  The real stream must not see it: same reconstructions, same delivery, same coded pictures."""
    W, H, N, enc_mode, tune, qp, intra_period = 256, 192, 36, 8, 1, 40, 19
    on = run_clip(W, H, N, enc_mode, tune, qp, 1, intra_period, False)
    off = run_clip(W, H, N, enc_mode, tune, qp, 1, intra_period, False, env={"SVT_HIP_WARMUP": "0"})
    for run in (on, off):
        _, recon, order, flags_seen, packets = run[:5]
        assert sorted(order) == list(range(N)) and len(packets) == N and packets[-1][1] & 1 and flags_seen[-1] == 1
    assert on[2] == off[2]
    for k in range(N):
        assert np.array_equal(on[1][k], off[1][k]), k
    for k in on[6]:   # the padded reference pictures and the coded pictures the library still holds at the end
        assert np.array_equal(on[6][k], off[6][k]), k
        assert np.array_equal(on[5][k]["q"], off[5][k]["q"]) and np.array_equal(on[5][k]["em"], off[5][k]["em"]), k


def test_eight_contexts_c4_shaped_clip_equals_one_context():
    """BASELINE configuration C4's host shape on one GPU (SURVEY section 7's determinism contract at C4's real GOP count): 520 pictures =
    8 closed GOPs of 65 (-intra-period 64), dealt to EIGHT contexts (`SVT_HIP_DEVICES=0,0,0,0,0,0,0,0`: GOP g on device g mod 8, each with
    its own picture ring, streams, workspace and feeder thread, as on an 8-GPU node) -- every reconstruction must equal the one-context
    run byte for byte, in the same delivery semantics (all pictures delivered, EOS on the last).  The first GOP is also checked against the
    oracle chain, so that "equal" cannot mean "equally wrong"."""
    W, H, N, enc_mode, tune, qp, intra_period = 136, 72, 520, 8, 1, 40, 64
    base = T.gen_clip_subpel(W, H, 30, 59)
    clip = [base[i % 30] if (i // 30) % 2 == 0 else base[29 - i % 30] for i in range(N)]
    one = run_clip(W, H, N, enc_mode, tune, qp, 1, intra_period, False, frames=clip)
    eight = run_clip(W, H, N, enc_mode, tune, qp, 1, intra_period, False, frames=clip, env={"SVT_HIP_DEVICES": "0,0,0,0,0,0,0,0"})
    for run in (one, eight):
        _, recon, order, flags_seen, packets = run[:5]
        assert sorted(order) == list(range(N)) and len(packets) == N and packets[-1][1] & 1 and flags_seen[-1] == 1
    for k in range(N):
        assert np.array_equal(one[1][k], eight[1][k]), k
    n0 = intra_period + 1
    recs, _ = oracle_clip(clip[:n0], W, H, n0, enc_mode, tune, qp, 1, intra_period, False)
    for k in range(n0):
        y, u, v = recs[k].interior()
        assert np.array_equal(eight[1][k], np.concatenate([y.ravel(), u.ravel(), v.ravel()])), k


def test_without_recon_output_only_reference_pictures_are_reconstructed():
    """recon_file = 0 at enc-mode 8: base-layer pictures deblocked, layers 1-3 reconstructed without deblocking (the reference allows
    the encoder / decoder mismatch there), the deepest layer not reconstructed at all -- and eb_vp9_svt_get_recon answers EB_ErrorMax"""
    W, H, N, enc_mode, tune, qp = 256, 192, 34, 8, 1, 40
    frames, recon, order, flags_seen, packets, infos, refpics, _ = run_clip(W, H, N, enc_mode, tune, qp, 0, -1, False, seed=47)
    recs, outs = oracle_clip(frames, W, H, N, enc_mode, tune, qp, 0, -1, False)
    assert not recon and len(packets) == N
    kinds = set()
    for k, d in infos.items():
        if outs[k]["intra"]:
            continue
        i, fl = d["info"], outs[k]["flags"]
        kinds.add((i.do_recon, i.apply_loop_filter, i.pad_reference))
        assert (i.do_recon, i.apply_loop_filter, i.pad_reference) == (fl.do_recon, fl.apply_loop_filter, fl.pad_reference)
        assert np.array_equal(d["q"], outs[k]["o"]["qcoeff"]), k
        if i.pad_reference:
            assert np.array_equal(refpics[k], recs[k].buf[:refpics[k].size]), k
    assert kinds == {(1, 1, 1), (1, 0, 1), (0, 0, 0)}


@pytest.mark.parametrize("env", [
    {"SVT_HIP_RING_GROUPS": "2"},                                                  # the smallest ring, feeder thread, upload / key streams (the defaults)
    {"SVT_HIP_RING_GROUPS": "2", "SVT_HIP_FEEDER": "0"},                           # everything enqueued by the caller's thread
    {"SVT_HIP_RING_GROUPS": "2", "SVT_HIP_DEEP_STREAM": "1"},                      # + the deep-layer stream
    {"SVT_HIP_RING_GROUPS": "3", "SVT_HIP_NO_KEY_STREAM": "1", "SVT_HIP_NO_UPLOAD_STREAM": "1"},
    {},                                                                            # the default ring of four mini-GOPs
], ids=["ring2", "ring2-nofeeder", "ring2-deep", "ring3-nokey-noupload", "default"])
def test_long_clip_reuses_every_picture_slot_several_times(env):
    """90 pictures with an intra refresh every 40: every slot of a 34-picture ring is overwritten two or three times while the upload, input,
    main, key and output streams run side by side and the device's feeder thread enqueues the groups -- a slot must not be overwritten before
    its last reader (the marker chain of the library) nor read before its upload, whatever the set of streams and threads.  Every
    reconstruction still equals the oracle chain."""
    W, H, N, enc_mode, tune, qp, intra_period = 136, 72, 90, 8, 1, 40, 39
    base = T.gen_clip_subpel(W, H, 30, 53)
    clip = [base[i % 30] if (i // 30) % 2 == 0 else base[29 - i % 30] for i in range(N)]          # 30 pictures forth and back
    frames, recon, order, flags_seen, packets, infos, refpics, _ = run_clip(W, H, N, enc_mode, tune, qp, 1, intra_period, False, frames=clip, env=env)
    recs, outs = oracle_clip(frames, W, H, N, enc_mode, tune, qp, 1, intra_period, False)
    assert sorted(order) == list(range(N)) and len(packets) == N and packets[-1][1] & 1 and flags_seen[-1] == 1
    for k in range(N):
        y, u, v = recs[k].interior()
        assert np.array_equal(recon[k], np.concatenate([y.ravel(), u.ravel(), v.ravel()])), k


def test_enc_mode_3_tune_0_through_the_api():
    """BASELINE C5's parameters behind the public API: tune 0 (SQ) -> 3 hierarchical levels, 8-picture mini-GOPs; enc-mode 3 -> the 64x64
    search area with the SSD sub-pel search on every PU (the C5 instances of the ME kernel), the SQ variant of the stage flags.  Every
    reconstruction equals the oracle chain."""
    W, H, N, enc_mode, tune, qp = 192, 136, 19, 3, 0, 36
    frames, recon, order, flags_seen, packets, infos, refpics, _ = run_clip(W, H, N, enc_mode, tune, qp, 1, -1, False, seed=61)
    recs, outs = oracle_clip(frames, W, H, N, enc_mode, tune, qp, 1, -1, False)
    assert sorted(order) == list(range(N)) and len(packets) == N
    for k in range(N):
        y, u, v = recs[k].interior()
        assert np.array_equal(recon[k], np.concatenate([y.ravel(), u.ravel(), v.ravel()])), (k, outs[k].get("layer"))
    levels = {d["info"].hierarchical_levels for k, d in infos.items() if not outs[k]["intra"]}
    assert levels <= {0, 1, 2, 3} and 3 in levels


@pytest.mark.parametrize("enc_mode,tune", [(5, 2), (9, 1), (0, 1), (6, 0)])
def test_other_presets_through_the_api(enc_mode, tune):
    """more corners of the preset table behind the API (VMAF tune, the fastest and the slowest enc-modes, SQ at enc-mode 6): the ME
    parameters, the mini-GOP structure and the stage flags the library derives must be the ones the oracle chain is run with"""
    W, H, N, qp = 136, 72, 20, 44
    frames, recon, order, flags_seen, packets, infos, refpics, _ = run_clip(W, H, N, enc_mode, tune, qp, 1, -1, False, seed=70 + enc_mode)
    recs, outs = oracle_clip(frames, W, H, N, enc_mode, tune, qp, 1, -1, False)
    assert sorted(order) == list(range(N)) and len(packets) == N
    for k in range(N):
        y, u, v = recs[k].interior()
        assert np.array_equal(recon[k], np.concatenate([y.ravel(), u.ravel(), v.ravel()])), (k, outs[k].get("layer"))


def test_malformed_callback_grid_ends_the_stream():
    """a mode-decision callback that hands back a grid the path cannot code (here: a 32x32 transform in an 8x8 block): the library
    checks the grid on the host before uploading it, the call that flushes the group fails, and every later call answers EB_ErrorMax --
    nothing is coded from the slot's previous contents, nothing is delivered as a success"""
    lib = shim()
    for f_ in ("svt_vp9_shim_set_mode_decision", "eb_vp9_svt_enc_send_picture", "eb_vp9_svt_get_packet"):
        getattr(lib, f_).restype = C.c_int32
    W, H, N = 192, 128, 18
    frames = T.gen_clip_subpel(W, H, N, 5)
    cfg, h = Cfg(), C.c_void_p()
    assert lib.eb_vp9_svt_init_handle(C.byref(h), None, C.byref(cfg)) == 0
    cfg.source_width, cfg.source_height, cfg.enc_mode, cfg.tune, cfg.frame_rate, cfg.intra_period, cfg.qp, cfg.recon_file = W, H, 8, 1, 60 << 16, -1, 40, 0
    assert lib.eb_vp9_svt_enc_set_parameter(h, C.byref(cfg)) == 0
    calls = []

    def cb(user, info, me_p, mc_p, lf_p, mi_stride):
        calls.append(int(info.contents.picture_number))
        if info.contents.is_intra:
            return 1
        lf = np.zeros((H // 8, W // 8), dtype=B.LF_MODE_INFO_DTYPE)
        lf["sb_type"], lf["tx_size"], lf["is_inter"] = 3, 3, 1          # 8x8 blocks with a 32x32 transform
        C.memmove(lf_p, lf.ctypes.data, lf.nbytes)
        return 0
    keep = MD_CB(cb)
    assert lib.svt_vp9_shim_set_mode_decision(h, keep, None) == 0
    assert lib.eb_vp9_init_encoder(h) == 0
    rcs = []
    for n in range(N):
        y = np.ascontiguousarray(frames[n])
        u, v = (np.ascontiguousarray(p) for p in chroma(y, n))
        i = In(y.ctypes.data, u.ctypes.data, v.ctypes.data, None, None, None, W, W // 2, W // 2)
        b = Hdr(size=C.sizeof(Hdr), p_buffer=C.addressof(i), pts=n, flags=1 if n == N - 1 else 0)
        rcs.append(lib.eb_vp9_svt_enc_send_picture(h, C.byref(b)) & 0xffffffff)
    bad = [k for k, rc in enumerate(rcs) if rc != 0]
    assert bad and calls, (rcs, calls)
    assert rcs[bad[0]] == 0x80001005 and all(rc == 0x7FFFFFFF for rc in rcs[bad[0] + 1:]), [hex(r) for r in rcs]   # EB_ErrorBadParameter, then EB_ErrorMax
    pp = C.POINTER(Hdr)()
    assert lib.eb_vp9_svt_get_packet(h, C.byref(pp), C.c_uint8(1)) & 0xffffffff == 0x7FFFFFFF
    assert lib.eb_vp9_deinit_encoder(h) == 0 and lib.eb_vp9_deinit_handle(h) == 0
