"""The oracle's view of the picture-level EncDec chain (test infrastructure): inter prediction -> transform / quantisation /
reconstruction over the blocks of the mode-info grid -> skip flags -> loop-filter masks -> deblocking -> reference padding,
each stage the oracle's (oracle/*.c), composed on the host exactly as svt_hip_encdec_batch_device composes the kernels.
Used by tests/test_gpu_encdec.py and tests/test_enc_shim.py to check the device chain and the encoder shim's reconstruction."""
import ctypes as C
import time

import numpy as np

import svt_testlib as T

B = T.B
PAD = 80
_W4 = [1, 1, 2, 2, 2, 4, 4, 4, 8, 8, 8, 16, 16]
_H4 = [1, 2, 1, 2, 4, 2, 4, 8, 4, 8, 16, 8, 16]


class RefPic:
    """a padded reference / reconstruction picture: three planes in one buffer (Y with `pad` samples of border, Cb and Cr with
    pad / 2), the layout the reference gives its reference pictures (Codec/EbEncHandle.c:968-971)"""

    def __init__(self, W, H, pad=PAD, fill=None):
        self.W, self.H, self.pad = W, H, pad
        self.pw, self.ph, self.cpw, self.cph = W + 2 * pad, H + 2 * pad, W // 2 + pad, H // 2 + pad
        self.u_base = self.pw * self.ph
        self.v_base = self.u_base + self.cpw * self.cph
        self.nbytes = (self.v_base + self.cpw * self.cph + 63) // 64 * 64
        self.buf = np.zeros(self.nbytes, np.uint8) if fill is None else np.full(self.nbytes, fill, np.uint8)

    def planes(self, buf=None):
        b = self.buf if buf is None else buf
        return (b[:self.u_base].reshape(self.ph, self.pw), b[self.u_base:self.v_base].reshape(self.cph, self.cpw),
                b[self.v_base:self.v_base + self.cpw * self.cph].reshape(self.cph, self.cpw))

    def offsets(self):
        """byte offsets of sample (0, 0) of Y, Cb, Cr"""
        p = self.pad
        return (p * self.pw + p, self.u_base + (p // 2) * self.cpw + p // 2, self.v_base + (p // 2) * self.cpw + p // 2)

    def interior(self, buf=None):
        y, u, v = self.planes(buf)
        p = self.pad
        return y[p:p + self.H, p:p + self.W], u[p // 2:p // 2 + self.H // 2, p // 2:p // 2 + self.W // 2], v[p // 2:p // 2 + self.H // 2, p // 2:p // 2 + self.W // 2]

    def set_padded(self, y, u, v):
        py, pu, pv = self.planes()
        p = self.pad
        py[:] = np.pad(y, p, mode="edge")
        pu[:] = np.pad(u, p // 2, mode="edge")
        pv[:] = np.pad(v, p // 2, mode="edge")
        return self

    def desc(self, base):
        d = B.YuvPlanes()
        o = self.offsets()
        d.y, d.u, d.v = base + o[0], base + o[1], base + o[2]
        d.y_stride, d.uv_stride, d.width, d.height = self.pw, self.cpw, self.W, self.H
        return d


def uv_tx(bs, tx):
    m = min(max(_W4[bs] // 2, 1), max(_H4[bs] // 2, 1))
    return min(tx, 3 if m >= 8 else 2 if m >= 4 else 1 if m >= 2 else 0)


def host_block_list(lf_mis, geoms, mi_stride):
    """the product's HOST list builder (tests/test_encdec_host.py checks it against an independent enumeration)"""
    n = len(lf_mis)
    W, H = geoms[0].width, geoms[0].height
    cap = n * W * H * 3 // 32
    arr = (C.c_void_p * n)(*[m.ctypes.data for m in lf_mis])
    gs = (B.TqPicGeom * n)(*geoms)
    blocks = np.zeros(cap, dtype=B.TQ_BLOCK_DTYPE)
    pos = np.zeros(cap, np.uint32)
    cnt = (C.c_int32 * 4)()
    rc = B.load().svt_hip_tq_blocks_from_grid(n, arr, mi_stride, gs, blocks.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p), cap, cnt)
    assert rc >= 0, rc
    return blocks[:rc].copy(), pos[:rc].copy(), list(cnt)


def qtabs_of(q_index):
    out = (B.QuantTables * 2)()
    assert B.load().svt_hip_quant_tables_for_qindex(q_index, out) == 0
    return np.frombuffer(bytes(out), dtype=B.QUANT_DTYPE).copy()


def eob_map_offsets(W, H):
    w4, h4 = W // 4, H // 4
    return 0, w4 * h4, w4 * h4 + (w4 // 2) * (h4 // 2), w4 * h4 + 2 * (w4 // 2) * (h4 // 2)


def oracle_encdec_picture(src, refs, mc_mi, lf_mi, q_index, flags, thr, use_subpel=1, pad=PAD, recon_init=None, timings=None):
    """src = (y, u, v) tight planes; refs = two RefPic; mc_mi / lf_mi [mi_rows][mi_cols] records.  Returns a dict with the padded
    reconstruction buffer (RefPic layout), qcoeff / dqcoeff in the driver's position-addressed layout, the eob map, the updated lf
    grid (skip flags) and the masks."""
    H, W = src[0].shape
    mi_rows, mi_cols = H // 8, W // 8
    mc_mi, lf_mi = np.ascontiguousarray(mc_mi), np.ascontiguousarray(lf_mi).copy()
    tm = timings if timings is not None else {}
    t0 = time.perf_counter()

    def lap(name):
        nonlocal t0
        t1 = time.perf_counter()
        tm[name] = tm.get(name, 0.0) + (t1 - t0)
        t0 = t1
    # 1. inter prediction
    mcase = dict(mi=mc_mi, mi_rows=mi_rows, mi_cols=mi_cols, refs=[r.planes() for r in refs], pad=pad, use_subpel=use_subpel, width=W, height=H)
    pred = T.oracle_mc_frame(mcase)
    lap("mc")
    # 2. the blocks: source and prediction as tight planes one after the other, the reconstruction padded
    rec = RefPic(W, H, pad) if recon_init is None else recon_init
    g = B.TqPicGeom()
    g.width, g.height = W, H
    so = (0, W * H, W * H + (W // 2) * (H // 2))
    ro = rec.offsets()
    for k in range(3):
        g.src_off[k] = g.pred_off[k] = so[k]
        g.recon_off[k] = ro[k]
    g.src_stride[0] = g.pred_stride[0] = W
    g.src_stride[1] = g.pred_stride[1] = W // 2
    g.recon_stride[0], g.recon_stride[1] = rec.pw, rec.cpw
    g.coeff_base, g.recon_set, g.pic, g.do_recon = 0, 0, 0, int(flags.do_recon)
    blocks, pos, cnt = host_block_list([lf_mi], [g], lf_mi.shape[1])
    blocks["pad"] &= 0x0F                                       # the oracle has one reconstruction buffer
    srcb = np.concatenate([p.ravel() for p in src])
    predb = np.concatenate([p.ravel() for p in pred])
    n_coeff = T.n_sb(W, H) * B.SB_COEFFS
    q, dq = np.zeros(n_coeff, np.int16), np.zeros(n_coeff, np.int16)
    eob = np.zeros(len(blocks), np.uint16)
    iscan, _ = T.iscan_array()
    # (the product's iscan offsets index ITS table, which has one 32x32 entry: rebase onto the test table's offsets)
    _, offs = T.iscan_array()
    blocks["iscan_off"] = [offs[(int(t), int(tt))] for t, tt in zip(blocks["tx_size"], blocks["tx_type"])]
    qt = qtabs_of(q_index)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lap("lists")
    rc = T.oracle().svt_oracle_tq_batch(vp(srcb), vp(predb), vp(rec.buf), vp(blocks), len(blocks), vp(qt), vp(iscan), vp(q), vp(dq), vp(eob))
    assert rc == 0
    lap("tq")
    # 3. eob map, skip flags
    e0, e1, e2, e3 = eob_map_offsets(W, H)
    emap = np.zeros(e3, np.uint16)
    nz = np.zeros((mi_rows, mi_cols), bool)
    intra_units = lf_mi["is_inter"][:, :mi_cols] == 0
    if intra_units.any():
        # intra blocks of an inter picture: coded behind the batch, from their neighbours' reconstruction (oracle/oracle_intra.c, mixed)
        assert flags.do_recon
        oi = oracle_intra_picture(src, lf_mi, q_index, rec=rec, pad=pad, mixed=1, pred_init=pred, q_init=q, dq_init=dq, emap_init=emap)
        pred = oi["pred"]
        ey = emap[e0:e1].reshape(H // 4, W // 4)
        eu, ev = emap[e1:e2].reshape(H // 8, W // 8), emap[e2:e3].reshape(H // 8, W // 8)
        nz |= intra_units & ((ey.reshape(H // 8, 2, W // 8, 2) != 0).any(axis=(1, 3)) | (eu != 0) | (ev != 0))     # set at the block's first unit
    plane = (pos >> 22) & 3
    y4, x4 = (pos >> 11) & 0x7FF, pos & 0x7FF
    pw4 = np.where(plane == 0, W // 4, W // 8)
    emap[np.array([e0, e1, e2])[plane] + y4 * pw4 + x4] = eob
    uy, ux = np.where(plane == 0, y4 >> 1, y4), np.where(plane == 0, x4 >> 1, x4)
    w8 = np.maximum(np.array(_W4)[lf_mi["sb_type"]] // 2, 1)
    h8 = np.maximum(np.array(_H4)[lf_mi["sb_type"]] // 2, 1)
    hit = eob > 0
    oy, ox = uy - uy % h8[uy, ux], ux - ux % w8[uy, ux]
    nz[oy[hit], ox[hit]] = True
    r, c = np.meshgrid(np.arange(mi_rows), np.arange(mi_cols), indexing="ij")
    lf_mi["skip"][:, :mi_cols] = ~nz[r - r % h8[:, :mi_cols], c - c % w8[:, :mi_cols]]
    lap("skip")
    out = dict(rec=rec, qcoeff=q, dqcoeff=dq, eob_map=emap, lf_mi=lf_mi, blocks=blocks, pos=pos, eob=eob, counts=cnt, pred=pred, lfm=None)
    # 4. deblocking
    if flags.apply_loop_filter:
        lfm = T.oracle_lf_build_masks(lf_mi, mi_rows, mi_cols)
        out["lfm"] = lfm
        d = rec.desc(rec.buf.ctypes.data)
        lfm_c = np.ascontiguousarray(lfm)
        rc = T.oracle().svt_oracle_lf_frame(C.byref(d), lfm_c.ctypes.data_as(C.c_void_p), lfm_c.shape[1], C.byref(thr), mi_rows, mi_cols, 0)
        assert rc == 0
    lap("lf")
    # 5. the border
    if flags.pad_reference:
        d = rec.desc(rec.buf.ctypes.data)
        assert T.oracle().svt_oracle_ref_pad(C.byref(d), pad, pad) == 0
    lap("pad")
    return out


# ---------------------------------------------------------------------------------------------------
# intra pictures
# ---------------------------------------------------------------------------------------------------
INTRA_TX_TYPE = (0, 1, 2, 0, 3, 1, 2, 2, 1, 3)   # eb_vp9_intra_mode_to_tx_type_lookup (VPX/vp9_reconintra.c:20-31)


def gen_intra_grid(seed, W, H, sizes=(8, 16, 32), modes=tuple(range(10)), filter_level=20, mi_stride=None):
    """a random intra partition of a W x H picture: per 32x32 area one 32x32 block, four 16x16 areas, each of which is one 16x16
    block or four 8x8 blocks (areas that cross the picture edge always split); random luma / chroma modes per block"""
    rng = np.random.default_rng(seed)
    mi_rows, mi_cols = H // 8, W // 8
    mi = np.zeros((mi_rows, mi_stride or mi_cols), dtype=B.LF_MODE_INFO_DTYPE)
    mi["sb_type"] = 255

    def put(r, c, n8):
        rec = mi[r:r + n8, c:c + n8]
        rec["sb_type"], rec["tx_size"] = {1: 3, 2: 6, 4: 9}[n8], {1: 1, 2: 2, 4: 3}[n8]
        rec["is_inter"], rec["skip"], rec["filter_level"] = 0, 0, filter_level
        pad = rec["pad"]
        pad[..., 0], pad[..., 1], pad[..., 2] = 0, rng.choice(modes), rng.choice(modes)
        if n8 == 1 and 4 in sizes and (8 not in sizes or rng.random() < 0.4):
            # an 8x8 unit of four 4x4 luma blocks (+ one 4x4 chroma block per plane): modes of blocks 0..3 in nibbles
            m4 = [int(rng.choice(modes)) for _ in range(4)]
            rec["sb_type"], rec["tx_size"] = 0, 0
            pad[..., 1], pad[..., 0] = m4[0] | m4[1] << 4, m4[2] | m4[3] << 4

    def split(r, c, n8):
        fits = r + n8 <= mi_rows and c + n8 <= mi_cols
        allowed = (8 * n8) in sizes or (n8 == 1 and 4 in sizes)
        smaller = any(s < 8 * n8 for s in sizes)
        if fits and allowed and (n8 == 1 or not smaller or rng.random() < 0.45):
            put(r, c, n8)
            return
        assert n8 > 1, "an 8x8 block must be allowed at the picture edge"
        h = n8 // 2
        for dr in (0, h):
            for dc in (0, h):
                if r + dr < mi_rows and c + dc < mi_cols:
                    split(r + dr, c + dc, h)
    for r in range(0, mi_rows, 4):
        for c in range(0, mi_cols, 4):
            split(r, c, 4)
    return mi


def oracle_intra_picture(src, lf_mi, q_index, rec=None, pad=PAD, mixed=0, pred_init=None, q_init=None, dq_init=None, emap_init=None):
    """the oracle's intra encode pass (oracle/oracle_intra.c) into a RefPic (or `rec`): prediction, coefficients, eob map, reconstruction
    before deblocking.  mixed = 1: the intra blocks of an inter picture whose inter blocks are already in rec / pred_init / q_init ..."""
    H, W = src[0].shape
    rec = RefPic(W, H, pad) if rec is None else rec
    srcb = np.concatenate([p.ravel() for p in src])
    predb = np.zeros_like(srcb) if pred_init is None else np.concatenate([p.ravel() for p in pred_init])
    n_coeff = T.n_sb(W, H) * B.SB_COEFFS
    q = np.zeros(n_coeff, np.int16) if q_init is None else q_init
    dq = np.zeros(n_coeff, np.int16) if dq_init is None else dq_init
    emap = np.zeros(eob_map_offsets(W, H)[3], np.uint16) if emap_init is None else emap_init
    iscan, offs = T.iscan_array()
    ioff = (C.c_uint32 * 16)(*[offs[(ts, tt)] for ts in range(4) for tt in range(4)])
    ro = (C.c_uint32 * 3)(*rec.offsets())
    rs = (C.c_int32 * 2)(rec.pw, rec.cpw)
    qt = qtabs_of(q_index)
    mi = np.ascontiguousarray(lf_mi)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = T.oracle().svt_oracle_intra_picture(vp(srcb), vp(predb), vp(rec.buf), ro, rs, vp(mi), mi.shape[1], W, H, vp(qt), vp(iscan), ioff, vp(q), vp(dq), vp(emap), mixed)
    assert rc == 0, rc
    ny, nc = W * H, W * H // 4
    pred = [predb[:ny].reshape(H, W), predb[ny:ny + nc].reshape(H // 2, W // 2), predb[ny + nc:].reshape(H // 2, W // 2)]
    return dict(rec=rec, pred=pred, qcoeff=q, dqcoeff=dq, eob_map=emap)


def ref_intra_picture(src, lf_mi, q_index):
    """the same picture through the reference's own functions (oracle/_ref/ref_intra); tight planes"""
    import os, struct, subprocess, tempfile
    H, W = src[0].shape
    mi = np.ascontiguousarray(lf_mi)
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        with open(req, "wb") as f:
            f.write(struct.pack("<5i", 0x4E495653, W, H, mi.shape[1], q_index))
            for p in src:
                f.write(np.ascontiguousarray(p).tobytes())
            f.write(mi.tobytes())
        subprocess.check_call([os.path.join(T.REF_DIR, "ref_intra"), req, rsp])
        raw = open(rsp, "rb").read()
    ny, nc, o = W * H, W * H // 4, 0
    out = {}
    for name in ("pred", "rec"):
        planes = []
        for n, shp in ((ny, (H, W)), (nc, (H // 2, W // 2)), (nc, (H // 2, W // 2))):
            planes.append(np.frombuffer(raw, np.uint8, n, o).reshape(shp).copy())
            o += n
        out[name] = planes
    n_coeff = T.n_sb(W, H) * B.SB_COEFFS
    out["qcoeff"] = np.frombuffer(raw, np.int16, n_coeff, o).copy(); o += 2 * n_coeff
    out["dqcoeff"] = np.frombuffer(raw, np.int16, n_coeff, o).copy(); o += 2 * n_coeff
    ne = eob_map_offsets(W, H)[3]
    out["eob_map"] = np.frombuffer(raw, np.uint16, ne, o).copy()
    assert o + 2 * ne == len(raw)
    return out


def oracle_intra_chain(src, lf_mi, q_index, flags, thr, pad=PAD, recon_init=None):
    """an intra picture through the whole encode pass: oracle_intra_picture, then skip flags, masks, deblocking and border as
    oracle_encdec_picture does for an inter picture"""
    H, W = src[0].shape
    mi_rows, mi_cols = H // 8, W // 8
    lf_mi = np.ascontiguousarray(lf_mi).copy()
    out = oracle_intra_picture(src, lf_mi, q_index, rec=recon_init, pad=pad)
    rec, emap = out["rec"], out["eob_map"]
    e0, e1, e2, e3 = eob_map_offsets(W, H)
    ey = emap[e0:e1].reshape(H // 4, W // 4)
    eu, ev = emap[e1:e2].reshape(H // 8, W // 8), emap[e2:e3].reshape(H // 8, W // 8)
    # a block's transform blocks start at its first unit (one per plane; four luma ones in a unit of 4x4 blocks)
    any_nz = (ey.reshape(H // 8, 2, W // 8, 2) != 0).any(axis=(1, 3)) | (eu != 0) | (ev != 0)   # (a unit of 4x4 blocks has four luma entries)
    w8 = np.maximum(np.array(_W4)[lf_mi["sb_type"][:, :mi_cols]] // 2, 1)
    r, c = np.meshgrid(np.arange(mi_rows), np.arange(mi_cols), indexing="ij")
    lf_mi["skip"][:, :mi_cols] = ~any_nz[r - r % w8, c - c % w8]
    out["lf_mi"], out["lfm"] = lf_mi, None
    if flags.apply_loop_filter:
        lfm = T.oracle_lf_build_masks(lf_mi, mi_rows, mi_cols)
        out["lfm"] = lfm
        d = rec.desc(rec.buf.ctypes.data)
        lfm_c = np.ascontiguousarray(lfm)
        assert T.oracle().svt_oracle_lf_frame(C.byref(d), lfm_c.ctypes.data_as(C.c_void_p), lfm_c.shape[1], C.byref(thr), mi_rows, mi_cols, 0) == 0
    if flags.pad_reference:
        d = rec.desc(rec.buf.ctypes.data)
        assert T.oracle().svt_oracle_ref_pad(C.byref(d), pad, pad) == 0
    return out


def make_mixed(seed, lf_mi, mc_mi=None, share=0.3, level=None):
    """turns a share of the square blocks (8x8 .. 32x32) of an inter grid into intra blocks with random modes: lf grid is_inter = 0,
    tx_size = the block's own, pad[1] / pad[2] = luma / chroma mode; mc grid (when given) ref_list[0] = -1 (no inter prediction)"""
    rng = np.random.default_rng(seed)
    lf = np.ascontiguousarray(lf_mi).copy()
    mc = None if mc_mi is None else np.ascontiguousarray(mc_mi).copy()
    mi_rows, mi_cols = lf.shape[0], lf.shape[1] if mc is None else mc.shape[1]
    n = 0
    for r in range(mi_rows):
        for c in range(mi_cols):
            bs = int(lf["sb_type"][r, c])
            if bs not in (3, 6, 9):
                continue
            w8 = {3: 1, 6: 2, 9: 4}[bs]
            if r % w8 or c % w8 or rng.random() >= share:
                continue
            blk = lf[r:r + w8, c:c + w8]
            blk["is_inter"], blk["tx_size"], blk["skip"] = 0, {3: 1, 6: 2, 9: 3}[bs], 0
            blk["pad"][..., 0], blk["pad"][..., 1], blk["pad"][..., 2] = 0, rng.integers(0, 10), rng.integers(0, 10)
            if w8 == 1 and rng.random() < 0.35:                      # a unit of four 4x4 luma blocks, a mode each
                m4 = rng.integers(0, 10, 4)
                blk["sb_type"], blk["tx_size"] = 0, 0
                blk["pad"][..., 1], blk["pad"][..., 0] = int(m4[0]) | int(m4[1]) << 4, int(m4[2]) | int(m4[3]) << 4
            if level is not None:
                blk["filter_level"] = level
            if mc is not None:
                m = mc[r:r + w8, c:c + w8]
                m["ref_list"][..., 0], m["ref_list"][..., 1] = -1, -1
                m["mv_row"], m["mv_col"] = 0, 0
            n += 1
    return lf, mc, n


def ref_intra_picture_mixed(src, lf_mi, q_index, inter_rec):
    """the intra blocks of an inter picture through the reference's functions: inter_rec = tight planes holding the reconstruction of
    the inter blocks (oracle/_ref/ref_intra, mixed request)"""
    import os, struct, subprocess, tempfile
    H, W = src[0].shape
    mi = np.ascontiguousarray(lf_mi)
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        with open(req, "wb") as f:
            f.write(struct.pack("<5i", 0x4E495653, W, H, mi.shape[1], q_index | 1 << 16))
            for p in src:
                f.write(np.ascontiguousarray(p).tobytes())
            f.write(mi.tobytes())
            for p in inter_rec:
                f.write(np.ascontiguousarray(p).tobytes())
        subprocess.check_call([os.path.join(T.REF_DIR, "ref_intra"), req, rsp])
        raw = open(rsp, "rb").read()
    ny, nc, o = W * H, W * H // 4, 0
    out = {}
    for name in ("pred", "rec"):
        planes = []
        for n, shp in ((ny, (H, W)), (nc, (H // 2, W // 2)), (nc, (H // 2, W // 2))):
            planes.append(np.frombuffer(raw, np.uint8, n, o).reshape(shp).copy())
            o += n
        out[name] = planes
    n_coeff = T.n_sb(W, H) * B.SB_COEFFS
    out["qcoeff"] = np.frombuffer(raw, np.int16, n_coeff, o).copy(); o += 2 * n_coeff
    out["dqcoeff"] = np.frombuffer(raw, np.int16, n_coeff, o).copy(); o += 2 * n_coeff
    out["eob_map"] = np.frombuffer(raw, np.uint16, eob_map_offsets(W, H)[3], o).copy()
    return out
