"""Shared helpers for tests/, bench.py and __graft_entry__.smoke(): synthetic clips, PA planes,
loading the product binding and the parity oracle (oracle/ is test infrastructure; see oracle/oracle_me.c)."""
import ctypes as C
import importlib.util
import os
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")


def load_binding():
    spec = importlib.util.spec_from_file_location("svtvp9_binding", os.path.join(ROOT, "svt-vp9_amd", "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


B = load_binding()

# ---------------------------------------------------------------------------------------------------
# synthetic input: translating texture + noise
# ---------------------------------------------------------------------------------------------------


def gen_clip(width, height, n_frames, seed, noise=3):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (height // 4 + 16 + 12, width // 4 + 16 + 12), dtype=np.uint8)
    big = np.repeat(np.repeat(base, 4, axis=0), 4, axis=1)
    frames = []
    for n in range(n_frames):
        dx, dy = (5 * n) % 48, (2 * n) % 48
        y = big[dy:dy + height, dx:dx + width].astype(np.int16)
        y = y + rng.integers(-noise, noise + 1, y.shape, dtype=np.int16)
        frames.append(np.clip(y, 0, 255).astype(np.uint8))
    return frames


def gen_yuv(width, height, seed):
    y = gen_clip(width, height, 1, seed)[0]
    u = (y[::2, ::2] // 2 + 32).astype(np.uint8)
    v = np.full_like(u, 128)
    return y, u, v


class PaPic:
    """Padded input + 1/4 + 1/16 point-decimated planes, as PictureAnalysis produces them
    (Codec/EbPictureAnalysisProcess.c:102-122, 5043-5066; padding Codec/EbEncHandle.c:1003-1026)."""

    def __init__(self, luma):
        self.luma = luma
        self.full = np.ascontiguousarray(np.pad(luma, 68, mode="edge"))
        self.quarter = np.ascontiguousarray(np.pad(luma[::2, ::2], 32, mode="edge"))
        self.sixteenth = np.ascontiguousarray(np.pad(luma[::4, ::4], 16, mode="edge"))

    def desc(self):
        d = B.PaPicture()
        d.full = B.plane_desc(self.full, 68, 68)
        d.quarter = B.plane_desc(self.quarter, 32, 32)
        d.sixteenth = B.plane_desc(self.sixteenth, 16, 16)
        return d

    def planes(self):
        return [(self.full, 68), (self.quarter, 32), (self.sixteenth, 16)]


def gen_shifted_pair(width, height, dx, dy, seed, noise=2):
    """(reference, current) with current[y][x] = reference[y - dy][x - dx] (+ noise): one global displacement of any size,
    e.g. far inside an outer HME region"""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, ((height + abs(dy)) // 4 + 2, (width + abs(dx)) // 4 + 2), dtype=np.uint8)
    big = np.repeat(np.repeat(base, 4, axis=0), 4, axis=1)
    oy, ox = max(dy, 0), max(dx, 0)
    out = []
    for (y0, x0) in ((oy, ox), (oy - dy, ox - dx)):
        f = big[y0:y0 + height, x0:x0 + width].astype(np.int16) + rng.integers(-noise, noise + 1, (height, width), dtype=np.int16)
        out.append(np.clip(f, 0, 255).astype(np.uint8))
    return out


def n_sb(width, height):
    return ((width + 63) // 64) * ((height + 63) // 64)


# ---------------------------------------------------------------------------------------------------
# oracle (CPU restatement) and reference harness
# ---------------------------------------------------------------------------------------------------
_oracle = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
    if os.path.isdir("/root/reference/Source"):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
        _oracle = C.CDLL(path)
    return _oracle


def ref_kernels():
    path = os.path.join(REF_DIR, "libsvtref_kernels.so")
    return C.CDLL(path) if os.path.exists(path) else None


def have_ref(name):
    return os.path.exists(os.path.join(REF_DIR, name))


def oracle_me_picture(cur, ref0, ref1, params, sb_begin=0, sb_end=-1):
    w, h = cur.luma.shape[1], cur.luma.shape[0]
    nsb = n_sb(w, h)
    res = np.zeros((nsb, 85), dtype=B.ME_RESULT_DTYPE)
    rcme = np.zeros(nsb, dtype=np.uint32)
    dc, d0 = cur.desc(), ref0.desc()
    d1 = ref1.desc() if ref1 is not None else None
    rc = oracle().svt_oracle_me_picture(C.byref(dc), C.byref(d0), C.byref(d1) if d1 is not None else None,
                                        C.byref(params), res.ctypes.data_as(C.c_void_p),
                                        rcme.ctypes.data_as(C.c_void_p), sb_begin, sb_end)
    assert rc == 0
    return res, rcme


def oracle_me_picture_mt(cur, ref0, ref1, params, threads=None):
    """svt_oracle_me_picture over the whole picture, SB ranges spread over host threads (ctypes drops the GIL; an SB's result
    depends on nothing outside the SB, and the oracle keeps its working state per call)."""
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or min(os.cpu_count() or 1, 64)
    w, h = cur.luma.shape[1], cur.luma.shape[0]
    nsb = n_sb(w, h)
    res = np.zeros((nsb, 85), dtype=B.ME_RESULT_DTYPE)
    rcme = np.zeros(nsb, dtype=np.uint32)
    dc, d0 = cur.desc(), ref0.desc()
    d1 = ref1.desc() if ref1 is not None else None
    step = max(1, (nsb + 4 * threads - 1) // (4 * threads))

    def run(b):
        rc = oracle().svt_oracle_me_picture(C.byref(dc), C.byref(d0), C.byref(d1) if d1 is not None else None, C.byref(params),
                                            res.ctypes.data_as(C.c_void_p), rcme.ctypes.data_as(C.c_void_p), b, min(nsb, b + step))
        assert rc == 0
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(run, range(0, nsb, step)))
    return res, rcme


def ref_me_picture(cur, ref0, ref1, params, sb_begin=0, sb_end=-1):
    """Run the REFERENCE's motion_estimate_sb through oracle/_ref/ref_me_sb (build container only)."""
    exe = os.path.join(REF_DIR, "ref_me_sb")
    w, h = cur.luma.shape[1], cur.luma.shape[0]
    nsb = n_sb(w, h)
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        with open(req, "wb") as f:
            f.write(struct.pack("<i", 0x454D5653))
            f.write(bytes(params))
            f.write(struct.pack("<ii", sb_begin, sb_end))
            for pic in (cur, ref0, ref1 if ref1 is not None else ref0):
                for arr, pad in pic.planes():
                    hh, ww = arr.shape
                    f.write(struct.pack("<6i", ww, pad, pad, ww - 2 * pad, hh - 2 * pad, arr.size))
                    f.write(arr.tobytes())
        subprocess.check_call([exe, req, rsp])
        raw = open(rsp, "rb").read()
    n = struct.unpack_from("<i", raw, 0)[0]
    assert n == nsb
    res = np.frombuffer(raw, dtype=B.ME_RESULT_DTYPE, count=nsb * 85, offset=4).reshape(nsb, 85).copy()
    rcme = np.frombuffer(raw, dtype=np.uint32, count=nsb, offset=4 + nsb * 85 * 40).copy()
    return res, rcme


def me_results_equal(a, b, num_lists):
    """Field-wise comparison of the DEFINED fields of two [n_sb][85] result arrays.
    (x/y_mv_l1 are stale context memory in the reference when only list 0 is searched;
    candidates beyond `total` are never written.)"""
    bad = []
    for f in ("x_mv_l0", "y_mv_l0", "total", "dist0", "dir0"):
        if not np.array_equal(a[f], b[f]):
            bad.append(f)
    if num_lists == 2:
        for f in ("x_mv_l1", "y_mv_l1", "dist1", "dir1"):
            if not np.array_equal(a[f], b[f]):
                bad.append(f)
        m = a["total"] == 3
        for f in ("dist2", "dir2"):
            if not np.array_equal(a[f][m], b[f][m]):
                bad.append(f)
    return bad


def gen_clip_subpel(width, height, n_frames, seed, noise=2):
    """Smooth texture moving by fractional, region-dependent displacements, so that half/quarter-pel
    refinement and bi-prediction actually win (the integer-translation clip never exercises them)."""
    rng = np.random.default_rng(seed)
    H, W = height + 160, width + 160
    base = rng.integers(0, 256, (H // 8 + 2, W // 8 + 2)).astype(np.float64)
    # separable bilinear upsample x8 -> smooth texture
    yi = np.arange(H) / 8.0
    xi = np.arange(W) / 8.0
    y0, x0 = np.floor(yi).astype(int), np.floor(xi).astype(int)
    fy, fx = (yi - y0)[:, None], (xi - x0)[None, :]
    tex = (base[y0][:, x0] * (1 - fy) * (1 - fx) + base[y0 + 1][:, x0] * fy * (1 - fx) +
           base[y0][:, x0 + 1] * (1 - fy) * fx + base[y0 + 1][:, x0 + 1] * fy * fx)
    tex += rng.normal(0, 6, tex.shape)
    frames = []
    for n in range(n_frames):
        out = np.empty((height, width), np.float64)
        # two regions with different (fractional) motion
        for (r0, r1, vx, vy) in ((0, height // 2, 1.75, 0.5), (height // 2, height, -2.25, 1.25)):
            sx, sy = 80 + vx * n, 80 + vy * n
            ix, iy = int(np.floor(sx)), int(np.floor(sy))
            ax, ay = sx - ix, sy - iy
            blk = lambda dy, dx: tex[iy + dy + r0:iy + dy + r1, ix + dx:ix + dx + width]
            out[r0:r1] = (blk(0, 0) * (1 - ay) * (1 - ax) + blk(1, 0) * ay * (1 - ax) +
                          blk(0, 1) * (1 - ay) * ax + blk(1, 1) * ay * ax)
        out += rng.integers(-noise, noise + 1, out.shape)
        frames.append(np.clip(np.rint(out), 0, 255).astype(np.uint8))
    return frames


# ---------------------------------------------------------------------------------------------------
# host emulation of the HIP ME kernel (tests/emu; debugging aid for the CPU suite, never the product)
# ---------------------------------------------------------------------------------------------------
_emu = None


def emu():
    global _emu
    if _emu is None:
        d = os.path.join(ROOT, "tests", "emu")
        so = os.path.join(d, "libme_emu.so")
        srcs = [os.path.join(d, "me_emu.c"), os.path.join(ROOT, "svt-vp9_amd", "csrc", "me_core.h"),
                os.path.join(ROOT, "svt-vp9_amd", "csrc", "me_layout.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-fPIC", "-shared", "-Wno-unused-function", "-o", so,
                                   srcs[0]])
        _emu = C.CDLL(so)
    return _emu


def emu_me_picture(cur, ref0, ref1, params, sb_begin=0, sb_end=-1):
    w, h = cur.luma.shape[1], cur.luma.shape[0]
    nsb = n_sb(w, h)
    res = np.zeros((nsb, 85), dtype=B.ME_RESULT_DTYPE)
    rcme = np.zeros(nsb, dtype=np.uint32)
    dc, d0 = cur.desc(), ref0.desc()
    d1 = ref1.desc() if ref1 is not None else None
    rc = emu().svt_emu_me_picture(C.byref(dc), C.byref(d0), C.byref(d1) if d1 is not None else None,
                                  C.byref(params), res.ctypes.data_as(C.c_void_p),
                                  rcme.ctypes.data_as(C.c_void_p), sb_begin, sb_end)
    assert rc == 0, rc
    return res, rcme


# ---------------------------------------------------------------------------------------------------
# transform / quantisation helpers
# ---------------------------------------------------------------------------------------------------
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
TX_N = {0: 4, 1: 8, 2: 16, 3: 32}


class _ScanOrder(C.Structure):
    _fields_ = [("scan", C.POINTER(C.c_int16)), ("iscan", C.POINTER(C.c_int16)), ("neighbors", C.POINTER(C.c_int16))]


def ref_scan_tables():
    """Read the reference's scan / iscan tables (VPX/vp9_scan.c) out of oracle/_ref/libsvtref_kernels.so."""
    lib = ref_kernels()
    so = (_ScanOrder * 4 * 4).in_dll(lib, "eb_vp9_scan_orders")  # [TX_SIZES][TX_TYPES]
    out = {}
    for ts in range(4):
        n = TX_N[ts] ** 2
        for tt in range(4):
            out[f"scan_{ts}_{tt}"] = np.ctypeslib.as_array(so[ts][tt].scan, (n,)).copy()
            out[f"iscan_{ts}_{tt}"] = np.ctypeslib.as_array(so[ts][tt].iscan, (n,)).copy()
    return out


_scan_cache = None


def scan_tables():
    """Scan tables as committed test data (tests/golden/vp9_scan_tables.npz; generated by tests/gen_golden.py)."""
    global _scan_cache
    if _scan_cache is None:
        _scan_cache = dict(np.load(os.path.join(GOLDEN_DIR, "vp9_scan_tables.npz")))
    return _scan_cache


def iscan_array():
    """All iscan tables concatenated + {(tx_size, tx_type): element offset}.  32x32 always uses tx_type 0."""
    t = scan_tables()
    parts, offs, pos = [], {}, 0
    for ts in range(4):
        for tt in range(4):
            a = t[f"iscan_{ts}_{tt if ts < 3 else 0}"]
            offs[(ts, tt)] = pos
            parts.append(a)
            pos += a.size
    return np.concatenate(parts).astype(np.int16), offs


def quant_table(q_dc, q_ac):
    """[DC, AC] quantiser rows as eb_vp9_init_quantizer derives them from the step sizes
    (VPX/vp9_quantize.c:182-265: invert_quant, zbin = ROUND_POWER_OF_TWO(qzbin_factor * q, 7), round = 48*q>>7).
    Test-input generator only: any table is a valid quantiser input, and the zero-bin factor used here (64 below step 148) is NOT
    the reference's (84, get_qzbin_factor :192-204) -- the golden fixtures were produced from these inputs, so it stays.  The
    tables of a real q index come from svt_hip_quant_tables_init, which is pinned against eb_vp9_init_quantizer for all 256
    indices (tests/test_tq_params.py) and is what bench.py uses."""
    rec = np.zeros((), dtype=B.QUANT_DTYPE)
    for i, q in enumerate((q_dc, q_ac)):
        t = 1 << 16
        l = 0
        tt = q
        while tt > 1:
            l += 1
            tt >>= 1
        m = 1 + (1 << (16 + l)) // q
        rec["quant"][i] = np.int16(np.uint16((m - t) & 0xFFFF).view(np.int16)) if False else np.array(m - t).astype(np.int16)
        rec["quant_shift"][i] = 1 << (16 - l)
        qzbin_factor = 84 if q_ac == 0 else (64 if q < 148 else 80)
        rec["zbin"][i] = (qzbin_factor * q + 64) >> 7
        rec["round"][i] = (48 * q) >> 7
        rec["dequant"][i] = q
    return rec


def _p16(a):
    return a.ctypes.data_as(C.POINTER(C.c_int16))


def ref_fwd_txfm(res, tx_size, tx_type, partial32=False):
    lib = ref_kernels()
    n = TX_N[tx_size]
    res = np.ascontiguousarray(res, dtype=np.int16)
    out = np.zeros(n * n, dtype=np.int16)
    if tx_size == 3:
        (lib.eb_vpx_partial_fdct32x32_c if partial32 else lib.eb_vp9_fdct32x32_c)(_p16(res), _p16(out), n)
    else:
        fn = {0: lib.eb_vp9_fht4x4_c, 1: lib.eb_vp9_fht8x8_c, 2: lib.eb_vp9_fht16x16_c}[tx_size]
        fn(_p16(res), _p16(out), n, tx_type)
    return out


def oracle_fwd_txfm(res, tx_size, tx_type, partial32=False):
    n = TX_N[tx_size]
    res = np.ascontiguousarray(res, dtype=np.int16)
    out = np.zeros(n * n, dtype=np.int16)
    oracle().oracle_fwd_txfm(_p16(res), n, _p16(out), tx_size, tx_type, int(partial32))
    return out


def ref_quantize(coeff, tx_size, tx_type, qrec):
    lib = ref_kernels()
    t = scan_tables()
    n = coeff.size
    scan = np.ascontiguousarray(t[f"scan_{tx_size}_{tx_type if tx_size < 3 else 0}"], dtype=np.int16)
    iscan = np.ascontiguousarray(t[f"iscan_{tx_size}_{tx_type if tx_size < 3 else 0}"], dtype=np.int16)
    q, dq = np.zeros(n, np.int16), np.zeros(n, np.int16)
    eob = C.c_uint16(0)
    f = lambda k: _p16(np.ascontiguousarray(qrec[k], dtype=np.int16))
    arrs = {k: np.ascontiguousarray(qrec[k], dtype=np.int16) for k in ("zbin", "round", "quant", "quant_shift", "dequant")}
    fn = lib.eb_vp9_quantize_b_32x32_c if tx_size == 3 else lib.eb_vp9_quantize_b_c
    fn(_p16(coeff), C.c_long(n), 0, _p16(arrs["zbin"]), _p16(arrs["round"]), _p16(arrs["quant"]), _p16(arrs["quant_shift"]),
       _p16(q), _p16(dq), _p16(arrs["dequant"]), C.byref(eob), _p16(scan), _p16(iscan))
    return q, dq, eob.value


def oracle_quantize(coeff, tx_size, tx_type, qrec):
    t = scan_tables()
    n = coeff.size
    iscan = np.ascontiguousarray(t[f"iscan_{tx_size}_{tx_type if tx_size < 3 else 0}"], dtype=np.int16)
    q, dq = np.zeros(n, np.int16), np.zeros(n, np.int16)
    eob = C.c_uint16(0)
    rec = np.ascontiguousarray(qrec)
    oracle().oracle_quantize(_p16(coeff), n, rec.ctypes.data_as(C.c_void_p), _p16(q), _p16(dq), C.byref(eob), _p16(iscan),
                             int(tx_size == 3))
    return q, dq, eob.value


def ref_inv_add(dq, pred, tx_size, tx_type, eob):
    """The reference's wrapper logic (VPX/vp9_idct.c:111-189) applied with its own *_add_c kernels."""
    lib = ref_kernels()
    n = TX_N[tx_size]
    dst = np.ascontiguousarray(pred, dtype=np.uint8).copy()
    p8 = dst.ctypes.data_as(C.POINTER(C.c_uint8))
    d = _p16(np.ascontiguousarray(dq, dtype=np.int16))
    if tx_type != 0 and tx_size < 3:
        fn = {0: lib.eb_vp9_iht4x4_16_add_c, 1: lib.eb_vp9_iht8x8_64_add_c, 2: lib.eb_vp9_iht16x16_256_add_c}[tx_size]
        fn(d, p8, n, tx_type)
    elif tx_size == 0:
        (lib.eb_vp9_idct4x4_16_add_c if eob > 1 else lib.eb_vp9_idct4x4_1_add_c)(d, p8, n)
    elif tx_size == 1:
        (lib.eb_vp9_idct8x8_1_add_c if eob == 1 else lib.eb_vp9_idct8x8_12_add_c if eob <= 12 else lib.eb_vp9_idct8x8_64_add_c)(d, p8, n)
    elif tx_size == 2:
        (lib.eb_vp9_idct16x16_1_add_c if eob == 1 else lib.eb_vp9_idct16x16_10_add_c if eob <= 10 else
         lib.eb_vp9_idct16x16_38_add_c if eob <= 38 else lib.eb_vp9_idct16x16_256_add_c)(d, p8, n)
    else:
        (lib.eb_vp9_idct32x32_1_add_c if eob == 1 else lib.eb_vp9_idct32x32_34_add_c if eob <= 34 else
         lib.eb_vp9_idct32x32_135_add_c if eob <= 135 else lib.eb_vp9_idct32x32_1024_add_c)(d, p8, n)
    return dst


def oracle_inv_add(dq, pred, tx_size, tx_type, eob):
    n = TX_N[tx_size]
    dst = np.ascontiguousarray(pred, dtype=np.uint8).copy()
    oracle().oracle_inv_txfm_add(_p16(np.ascontiguousarray(dq, dtype=np.int16)), dst.ctypes.data_as(C.POINTER(C.c_uint8)), n,
                                 tx_size, tx_type, int(eob))
    return dst


def make_tq_case(seed, width=256, height=128, do_recon=True, qsteps=((40, 48), (8, 9), (200, 260)), extreme=False):
    """A plane pair (source, prediction) tiled with transform blocks of every size / type, grouped by size as the
    C ABI requires.  Returns dict(src, pred, blocks, counts, qtabs, iscan, n_coeff)."""
    rng = np.random.default_rng(seed)
    src = gen_clip(width, height, 1, seed)[0]
    if extreme:
        pred = np.where(rng.integers(0, 2, src.shape) > 0, 255, 0).astype(np.uint8)
        src = (255 - pred).astype(np.uint8) if seed % 2 else src
    else:
        pred = np.clip(np.roll(src, (1, 2), (0, 1)).astype(np.int16) + rng.integers(-12, 13, src.shape), 0, 255).astype(np.uint8)
        flat = rng.integers(0, 4, (height // 32, width // 32))
        for by in range(height // 32):     # some blocks with tiny residual (eob 0/1, reduced-eob inverse variants)
            for bx in range(width // 32):
                if flat[by, bx] == 0:
                    pred[by * 32:by * 32 + 32, bx * 32:bx * 32 + 32] = src[by * 32:by * 32 + 32, bx * 32:bx * 32 + 32]
                elif flat[by, bx] == 1:
                    pred[by * 32:by * 32 + 32, bx * 32:bx * 32 + 32] = np.clip(
                        src[by * 32:by * 32 + 32, bx * 32:bx * 32 + 32].astype(np.int16) + rng.integers(-2, 3), 0, 255)
    iscan, offs = iscan_array()
    qtabs = np.array([quant_table(a, b) for a, b in qsteps], dtype=B.QUANT_DTYPE)
    blocks = []
    # quadrant layout: every 32x32 area is assigned one transform size
    for by in range(height // 32):
        for bx in range(width // 32):
            ts = int(rng.integers(0, 4))
            n = TX_N[ts]
            for yy in range(0, 32, n):
                for xx in range(0, 32, n):
                    tt = int(rng.integers(0, 4)) if ts < 3 else 0
                    blocks.append((ts, tt, by * 32 + yy, bx * 32 + xx, int(rng.integers(0, len(qsteps))),
                                   int(ts == 3 and rng.integers(0, 4) == 0)))
    blocks.sort(key=lambda b: b[0])
    arr = np.zeros(len(blocks), dtype=B.TQ_BLOCK_DTYPE)
    pos = 0
    for i, (ts, tt, y, x, qi, part) in enumerate(blocks):
        n = TX_N[ts]
        arr[i] = (y * width + x, y * width + x, y * width + x, pos, offs[(ts, tt)], width, width, width, ts, tt, qi,
                  int(do_recon), part, 0)
        pos += n * n
    counts = np.array([sum(1 for b in blocks if b[0] == s) for s in range(4)], dtype=np.int32)
    return dict(src=src, pred=pred, blocks=arr, counts=counts, qtabs=qtabs, iscan=iscan, n_coeff=pos)


def make_tq_md_case(seed, width, height, n_cand):
    """The (block x candidate) form of mode decision's full loop: every transform block of the partition is coded n_cand
    times, each time against another prediction (same src_off, n_cand different pred_off -- the candidates' predictions are
    planes stacked below each other), nothing reconstructs (do_recon = 0)."""
    case = make_tq_case(seed, width=width, height=height, do_recon=False)
    rng = np.random.default_rng(seed + 77)
    src = case["src"]
    preds = [case["pred"]]
    for k in range(1, n_cand):
        preds.append(np.clip(np.roll(src, (k, -k), (0, 1)).astype(np.int16) + rng.integers(-6 * k, 6 * k + 1, src.shape), 0, 255).astype(np.uint8))
    base = case["blocks"]
    blocks = np.repeat(base, n_cand)                       # candidates of a block are neighbours in the list (grouping by size stays)
    cand = np.tile(np.arange(n_cand, dtype=np.uint32), len(base))
    blocks["pred_off"] += cand * np.uint32(src.size)
    blocks["recon_off"] = 0
    nn = 16 << (2 * blocks["tx_size"].astype(np.int64))
    blocks["coeff_off"] = np.concatenate([[0], np.cumsum(nn)[:-1]]).astype(np.uint32)
    return dict(src=src, pred=np.ascontiguousarray(np.concatenate(preds, axis=0)), blocks=blocks, counts=case["counts"] * n_cand, qtabs=case["qtabs"],
                iscan=case["iscan"], n_coeff=int(nn.sum()), n_cand=n_cand)


def oracle_tq_batch(case):
    recon = np.zeros_like(case["src"])
    q = np.zeros(case["n_coeff"], np.int16)
    dq = np.zeros(case["n_coeff"], np.int16)
    eob = np.zeros(len(case["blocks"]), np.uint16)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = oracle().svt_oracle_tq_batch(vp(case["src"]), vp(case["pred"]), vp(recon), vp(case["blocks"]), len(case["blocks"]),
                                      vp(case["qtabs"]), vp(case["iscan"]), vp(q), vp(dq), vp(eob))
    assert rc == 0
    return recon, q, dq, eob


def hip_tq_batch(ctx, case):
    lib = B.load()
    recon = np.zeros_like(case["src"])
    q = np.zeros(case["n_coeff"], np.int16)
    dq = np.zeros(case["n_coeff"], np.int16)
    eob = np.zeros(len(case["blocks"]), np.uint16)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    B.check(lib.svt_hip_tq_batch(ctx, vp(case["src"]), vp(case["pred"]), vp(recon), C.c_size_t(case["src"].size),
                                 vp(case["blocks"]), len(case["blocks"]), vp(case["qtabs"]), len(case["qtabs"]),
                                 vp(case["iscan"]), C.c_size_t(case["iscan"].size), vp(q), vp(dq),
                                 C.c_size_t(case["n_coeff"]), vp(eob)))
    return recon, q, dq, eob


# ---------------------------------------------------------------------------------------------------
# deblocking helpers
# ---------------------------------------------------------------------------------------------------
_LEFT_TX = {0: 0xFFFFFFFFFFFFFFFF, 1: 0xFFFFFFFFFFFFFFFF, 2: 0x5555555555555555, 3: 0x1111111111111111}
_ABOVE_TX = {0: 0xFFFFFFFFFFFFFFFF, 1: 0xFFFFFFFFFFFFFFFF, 2: 0x00FF00FF00FF00FF, 3: 0x000000FF000000FF}
_LEFT_TX_UV = {0: 0xFFFF, 1: 0xFFFF, 2: 0x5555, 3: 0x1111}
_ABOVE_TX_UV = {0: 0xFFFF, 1: 0xFFFF, 2: 0x0F0F, 3: 0x000F}


def _rect_mask(r, c, h, w, cols):
    m = 0
    for y in range(r, r + h):
        for x in range(c, c + w):
            m |= 1 << (y * cols + x)
    return m


def gen_lf_masks(rng, sb_rows, sb_cols, level_choices=(0, 8, 20, 33, 63)):
    """Random but well-formed LOOP_FILTER_MASKs (one per SB): a random quad-tree of prediction blocks per SB, each
    with a transform size <= its size, a skip flag and a filter level -- the bits are set with the same rules the
    reference's eb_vp9_build_mask uses (VPX/vp9_loopfilter.c:1587-1689), so no position carries two filter widths."""
    lfm = np.zeros((sb_rows, sb_cols), dtype=B.LF_MASK_DTYPE)
    for sr in range(sb_rows):
        for sc in range(sb_cols):
            left, above, i4 = [0, 0, 0, 0], [0, 0, 0, 0], 0
            lfl = np.zeros(64, np.uint8)

            def leaf(r, c, n, masks, tx_left, tx_above, cols, i4acc, maxtx):
                tx = int(rng.integers(0, min(maxtx, {1: 1, 2: 2, 4: 3, 8: 3}[n]) + 1))
                skip_inter = rng.integers(0, 3) == 0
                lvl = int(rng.choice(level_choices))
                if lvl == 0:
                    return 0, lvl
                size = _rect_mask(r, c, n, n, cols)
                masks[0][tx] |= _rect_mask(r, c, n, 1, cols)      # left edge of the prediction block
                masks[1][tx] |= _rect_mask(r, c, 1, n, cols)      # above edge
                if not skip_inter:
                    masks[0][tx] |= size & tx_left[tx]
                    masks[1][tx] |= size & tx_above[tx]
                    if tx == 0:
                        i4acc[0] |= size
                return size, lvl

            def split(r, c, n, fn):
                if n > 1 and rng.integers(0, 3) > 0:
                    h = n // 2
                    for dr in (0, h):
                        for dc in (0, h):
                            split(r + dr, c + dc, h, fn)
                else:
                    fn(r, c, n)

            acc_y = [0]
            my = ([0, 0, 0, 0], [0, 0, 0, 0])

            def fy(r, c, n):
                size, lvl = leaf(r, c, n, my, _LEFT_TX, _ABOVE_TX, 8, acc_y, 3)
                for y in range(r, r + n):
                    lfl[y * 8 + c:y * 8 + c + n] = lvl

            split(0, 0, 8, fy)
            acc_uv = [0]
            muv = ([0, 0, 0, 0], [0, 0, 0, 0])

            def fuv(r, c, n):
                # chroma filtering needs a non-zero level at the co-located luma position
                if lfl[(2 * r) * 8 + 2 * c] == 0:
                    return
                leaf(r, c, n, muv, _LEFT_TX_UV, _ABOVE_TX_UV, 4, acc_uv, 3)

            split(0, 0, 4, fuv)
            # a filter may only sit where the level is non-zero (the reference skips blocks with level 0)
            nz = 0
            for i in range(64):
                if lfl[i]:
                    nz |= 1 << i
            rec = lfm[sr, sc]
            for t in range(4):
                rec["left_y"][t] = my[0][t] & nz
                rec["above_y"][t] = my[1][t] & nz
                rec["left_uv"][t] = muv[0][t]
                rec["above_uv"][t] = muv[1][t]
            rec["int_4x4_y"] = acc_y[0] & nz
            rec["int_4x4_uv"] = acc_uv[0]
            rec["lfl_y"] = lfl
    return lfm


def make_lf_case(seed, width, height, sharpness=0):
    rng = np.random.default_rng(seed)
    y, u, v = gen_yuv(width, height, seed)
    # blocky reconstruction: quantise 8x8 / 16x16 means so that edges really get filtered (incl. flat / flat2 paths)
    hp, wp = (height + 15) // 16 * 16, (width + 15) // 16 * 16
    yb = np.pad(y.astype(np.int32), ((0, hp - height), (0, wp - width)), mode="edge")
    for n, frac in ((16, 0.5), (8, 0.5)):
        m = yb.reshape(hp // n, n, wp // n, n).mean(axis=(1, 3), keepdims=True)
        yb = np.where(rng.random((hp // n, 1, wp // n, 1)) < frac, m + rng.integers(-2, 3, (hp // n, n, wp // n, n)),
                      yb.reshape(hp // n, n, wp // n, n)).reshape(hp, wp)
    yb = yb[:height, :width]
    y = np.clip(yb, 0, 255).astype(np.uint8)
    u = np.clip((u.astype(np.int32) // 4) * 4 + rng.integers(-1, 2, u.shape), 0, 255).astype(np.uint8)
    v = np.clip(128 + rng.integers(-3, 4, v.shape) + (np.arange(v.shape[1]) // 8 % 2) * 6, 0, 255).astype(np.uint8)
    mi_rows, mi_cols = height // 8, width // 8
    sb_rows, sb_cols = (mi_rows + 7) // 8, (mi_cols + 7) // 8
    lfm = gen_lf_masks(rng, sb_rows, sb_cols)
    thr = B.LfThresh()
    oracle().svt_oracle_lf_thresh_init(C.byref(thr), sharpness)
    return dict(y=y, u=u, v=v, lfm=lfm, thr=thr, mi_rows=mi_rows, mi_cols=mi_cols, sharpness=sharpness)


def _yuv_desc(y, u, v):
    d = B.YuvPlanes()
    d.y, d.u, d.v = y.ctypes.data, u.ctypes.data, v.ctypes.data
    d.y_stride, d.uv_stride = y.strides[0], u.strides[0]
    d.width, d.height = y.shape[1], y.shape[0]
    return d


def oracle_lf_frame(case, y_only=False):
    y, u, v = case["y"].copy(), case["u"].copy(), case["v"].copy()
    d = _yuv_desc(y, u, v)
    lfm = np.ascontiguousarray(case["lfm"])
    rc = oracle().svt_oracle_lf_frame(C.byref(d), lfm.ctypes.data_as(C.c_void_p), lfm.shape[1], C.byref(case["thr"]),
                                      case["mi_rows"], case["mi_cols"], int(y_only))
    assert rc == 0
    return y, u, v


def ref_lf_frame(case, y_only=False):
    exe = os.path.join(REF_DIR, "ref_lf_frame")
    # the reference filters whole 8-sample groups and relies on the recon buffer's padding where a chroma block is
    # only 4 samples wide/high: give it padded planes and crop afterwards
    H0, W0 = case["y"].shape
    y = np.ascontiguousarray(np.pad(case["y"], ((0, 32), (0, 32)), mode="edge"))
    u = np.ascontiguousarray(np.pad(case["u"], ((0, 16), (0, 16)), mode="edge"))
    v = np.ascontiguousarray(np.pad(case["v"], ((0, 16), (0, 16)), mode="edge"))
    lfm = np.ascontiguousarray(case["lfm"])
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        with open(req, "wb") as f:
            f.write(struct.pack("<10i", 0x464C5653, y.shape[1], y.shape[0], y.shape[1], u.shape[1], case["mi_rows"],
                                case["mi_cols"], lfm.shape[1], lfm.size, int(y_only) | case["sharpness"] << 8))
            f.write(bytes(case["thr"]))
            f.write(lfm.tobytes())
            f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())
        subprocess.check_call([exe, req, rsp])
        raw = open(rsp, "rb").read()
    ys, us = y.size, u.size
    return (np.frombuffer(raw, np.uint8, ys).reshape(y.shape)[:H0, :W0].copy(),
            np.frombuffer(raw, np.uint8, us, ys).reshape(u.shape)[:H0 // 2, :W0 // 2].copy(),
            np.frombuffer(raw, np.uint8, us, ys + us).reshape(v.shape)[:H0 // 2, :W0 // 2].copy())


# ---- public encoder ABI (row b-1) ----------------------------------------------------------------------------------
API_VERIFY_CASES = (
    "", "enc_mode=8 tune=1", "source_width=3840 source_height=2160 enc_mode=8 tune=1", "source_width=3840 source_height=2160 enc_mode=11 tune=1",
    "source_width=3840 source_height=2160 enc_mode=12 tune=0", "source_width=3840 source_height=2160 enc_mode=13 tune=0", "enc_mode=10", "enc_mode=11",
    "source_width=640 source_height=360 enc_mode=9", "source_width=640 source_height=360 enc_mode=10", "source_width=60", "source_height=56",
    "source_width=1002", "source_width=1004", "source_height=1082", "source_width=8200", "source_height=4328", "source_width=8192 source_height=4320",
    "pred_structure=1", "pred_structure=0 base_layer_switch_mode=1", "base_layer_switch_mode=2", "qp=63", "qp=64", "intra_period=-2", "intra_period=-3",
    "intra_period=255", "intra_period=256", "loop_filter=2", "use_default_me_hme=2", "enable_hme_flag=2", "search_area_width=0", "search_area_width=256",
    "search_area_width=257", "search_area_height=0", "search_area_height=257", "level=10", "level=40", "level=41", "level=13", "level=62", "level=51 frame_rate=7864320",
    "source_width=3840 source_height=2160 level=50", "source_width=3840 source_height=2160 level=51 frame_rate=3932160", "frame_rate=0", "frame_rate=15728640", "frame_rate=15728641",
    "rate_control_mode=2", "rate_control_mode=3", "rate_control_mode=1 tune=1", "rate_control_mode=1 tune=0", "rate_control_mode=1 tune=0 max_qp_allowed=64",
    "rate_control_mode=0 max_qp_allowed=64", "rate_control_mode=2 min_qp_allowed=63", "rate_control_mode=2 min_qp_allowed=40 max_qp_allowed=30", "tune=2", "tune=3",
    "encoder_bit_depth=10", "profile=1", "speed_control_flag=2", "asm_type=2", "asm_type=0", "target_socket=1", "target_socket=2", "target_socket=-2",
)


def ref_api():
    """what oracle/_ref/ref_api reports about the reference's public API: layout lines, default configuration bytes, level
    tables, verify_settings' verdict for API_VERIFY_CASES"""
    exe = os.path.join(REF_DIR, "ref_api")
    run = lambda mode, inp=None: subprocess.run([exe, mode], input=inp, capture_output=True, check=True)
    layout = [ln for ln in run("layout").stdout.decode().splitlines() if ln.split()[0].split(".")[0] in
              ("EbComponentType", "EbSvtEncInput", "EbBufferHeaderType", "EbSvtVp9EncConfiguration", "enum")]
    grab = lambda r, tag: [ln for ln in r.stderr.decode().splitlines() if ln.startswith(tag)]
    defaults = np.array(grab(run("defaults"), "DEFAULTS")[0].split()[1:], np.int64).astype(np.uint8)
    levels = np.array(grab(run("levels"), "LEVELS")[0].split()[1:], np.uint64)
    verify = np.array([int(ln.split()[1]) for ln in grab(run("verify", "\n".join(API_VERIFY_CASES).encode() + b"\n"), "VERIFY")], np.int64)
    assert len(verify) == len(API_VERIFY_CASES)
    return dict(layout=np.array(layout), defaults=defaults, levels=levels, verify=verify)


# ---- stand-alone exhaustive SAD search (row M1) ----------------------------------------------------------------------
def make_sad_loop_case(seed, n_jobs=48):
    """Jobs shaped like the reference's three HME uses of eb_vp9_sad_loop_kernel (16x8 / 32x16 / 64x32 blocks whose rows are
    every other row of the plane: ref_stride = 2 x ref_stride_raw) over windows 1..224 wide, half of them on low-entropy
    data so that many positions tie (the first minimum in raster order must win)."""
    rng = np.random.default_rng(seed)
    src = np.zeros((64, 4096), np.uint8)
    ref = np.zeros((320, 1024), np.uint8)
    ref[:, :512] = rng.integers(0, 256, (320, 512))
    ref[:, 512:] = rng.integers(0, 3, (320, 512))
    jobs = np.zeros(n_jobs, dtype=B.SAD_LOOP_JOB_DTYPE)
    for j in range(n_jobs):
        bw, bh = ((16, 8), (32, 16), (64, 32))[j % 3]
        low = (j // 3) % 2
        sw = int(rng.integers(1, 225 if bw == 16 else 40)); sh = int(rng.integers(1, 113 if bw == 16 else 20))
        if j == 0: sw, sh = 224, 112
        if j == 1: sw, sh = 1, 1
        x0 = int(rng.integers(0, 512 - sw - bw)) + 512 * low
        y0 = int(rng.integers(0, 320 - sh - 2 * bh))
        blk = rng.integers(0, 3 if low else 256, (bh, bw))
        if j % 5 == 0:   # plant the block so that an exact match exists
            yy, xx = y0 + int(rng.integers(0, sh)), x0 + int(rng.integers(0, sw))
            blk = ref[yy:yy + 2 * bh:2, xx:xx + bw]
        src[:bh, 64 * j:64 * j + bw] = blk
        jobs[j] = (64 * j, y0 * 1024 + x0, 4096, 2048, 1024, bw, bh, sw, sh)
    return dict(src=src, ref=ref, jobs=jobs)


def _sad_loop_run(fn, case, is_ref):
    out = np.zeros((len(case["jobs"]), 3), np.int64)
    sp, rp = case["src"].ctypes.data, case["ref"].ctypes.data
    u8p = C.POINTER(C.c_uint8)
    for i, j in enumerate(case["jobs"]):
        best, x, y = C.c_uint64(0), C.c_int16(-1), C.c_int16(-1)
        sw, sh = int(j["search_w"]), int(j["search_h"])
        fn(C.cast(sp + int(j["src_off"]), u8p), int(j["src_stride"]), C.cast(rp + int(j["ref_off"]), u8p), int(j["ref_stride"]),
           int(j["height"]), int(j["width"]), C.byref(best), C.byref(x), C.byref(y), int(j["ref_stride_raw"]),
           C.c_int16(sw) if is_ref else sw, C.c_int16(sh) if is_ref else sh)
        out[i] = (best.value, x.value, y.value)
    return out


def oracle_sad_loop_case(case):
    return _sad_loop_run(oracle().oracle_sad_loop, case, False)


def ref_sad_loop_case(case):
    return _sad_loop_run(ref_kernels().eb_vp9_sad_loop_kernel, case, True)


def hip_sad_loop_case(ctx, case):
    import torch
    dev = torch.device("cuda", 0)
    ts, tr = torch.from_numpy(case["src"]).to(dev), torch.from_numpy(case["ref"]).to(dev)
    tj = torch.from_numpy(case["jobs"].view(np.uint8)).to(dev)
    n = len(case["jobs"])
    to = torch.zeros(n * 8, dtype=torch.uint8, device=dev)
    lib = B.load()
    B.check(lib.svt_hip_sad_loop_batch_device(ctx, C.c_void_p(ts.data_ptr()), C.c_void_p(tr.data_ptr()), C.c_void_p(tj.data_ptr()), n,
                                              C.c_void_p(to.data_ptr())))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    r = to.cpu().numpy().view(B.SAD_LOOP_RESULT_DTYPE)
    return np.stack([r["best_sad"].astype(np.int64), r["x"].astype(np.int64), r["y"].astype(np.int64)], axis=1)


# (W, H): every resolution class incl. the boundaries of eb_vp9_derive_input_resolution and the aspect-ratio split of the
# 1080i range; the sizes ADVICE r1 names (2048x1080, 1440x1080, 1600x900, 960x540, 1024x576) are among them
ME_PRESET_SIZES = ((640, 360), (720, 576), (960, 540), (1024, 576), (1000, 750), (1280, 720), (1600, 900), (1440, 1080), (2048, 512),
                   (4096, 256), (1536, 512), (1920, 1080), (2048, 1080), (1672, 1046), (2560, 1080), (2560, 1440), (3840, 2160), (4096, 2176),
                   (8192, 4320), (64, 64))
ME_PRESET_FIELDS = ("input_resolution", "enable_hme_flag", "enable_hme_level_0_flag", "enable_hme_level_1_flag", "enable_hme_level_2_flag",
                    "use_subpel", "cu8x8_mode", "cu16x16_mode", "single_hme_quadrant", "fractional_search_method", "fractional_search64x64",
                    "fractional_search_model", "search_area_width", "search_area_height", "number_hme_search_region_in_width",
                    "number_hme_search_region_in_height", "hme_level0_total_search_area_width", "hme_level0_total_search_area_height")
ME_PRESET_ARRAYS = ("hme_level0_search_area_in_width_array", "hme_level0_search_area_in_height_array", "hme_level1_search_area_in_width_array",
                    "hme_level1_search_area_in_height_array", "hme_level2_search_area_in_width_array", "hme_level2_search_area_in_height_array")


def me_preset_requests():
    """(W, H, enc_mode, tune, temporal_layer, is_used_as_reference, frame_rate) rows: every size class x tune x mode x layer."""
    rows = []
    for (w, h) in ME_PRESET_SIZES:
        for tune in range(3):
            for mode in range(13):
                for (tl, used) in ((0, 1), (1, 1), (3, 1), (4, 0)):
                    for fps in (60, 30):
                        rows.append((w, h, mode, tune, tl, used, fps))
    return np.array(rows, np.int32)


def ref_me_presets():
    """The reference's own derivation (oracle/_ref/ref_me_presets) for me_preset_requests(): dict(req, out int32 [n][30])."""
    req = me_preset_requests()
    text = "".join(" ".join(map(str, r)) + "\n" for r in req.tolist())
    out = subprocess.check_output([os.path.join(REF_DIR, "ref_me_presets")], input=text.encode()).decode()
    out = np.array([[int(x) for x in ln.split()] for ln in out.strip().splitlines()], np.int32)
    assert out.shape == (len(req), 30)
    return dict(req=req, out=out)


def product_me_preset_row(w, h, mode, tune, tl, used, fps):
    """svt_hip_me_params_derive + svt_hip_input_resolution in the column order of ref_me_presets() (use_subpel is the one
    column svt_me_params does not carry directly: it is fractional_search_model != 2)."""
    p = B.me_params_derive(pic_width=w, pic_height=h, enc_mode=mode, tune=tune, frame_rate=fps, num_ref_lists=2,
                           temporal_layer_index=tl, hierarchical_levels=4, is_used_as_reference=used)
    row = []
    for n in ME_PRESET_FIELDS:
        if n == "input_resolution":
            row.append(B.load().svt_hip_input_resolution(w, h))
        elif n == "use_subpel":
            row.append(int(p.fractional_search_model != 2))
        else:
            row.append(int(getattr(p, n)))
    for n in ME_PRESET_ARRAYS:
        row += [int(x) for x in getattr(p, n)]
    return row


def ref_lf_params():
    """The reference's own eb_vp9_loop_filter_init tables for sharpness 0..7 and eb_vp9_pick_filter_level's choice for
    every base_qindex (inter / key frame): dict(thr uint8 [8][3][64] = mblim, lim, hev_thr; pick int32 [2][256][3] =
    ac_quant(qindex), filter_level, sharpness_level)."""
    exe = os.path.join(REF_DIR, "ref_lf_frame")
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        open(req, "wb").write(struct.pack("<5i", 0x504C5653, 0, 0, 0, 0))
        subprocess.check_call([exe, req, rsp])
        raw = open(rsp, "rb").read()
    return dict(thr=np.frombuffer(raw, np.uint8, 8 * 192).reshape(8, 3, 64).copy(),
                pick=np.frombuffer(raw, np.int32, 2 * 256 * 3, 8 * 192).reshape(2, 256, 3).copy())


def hip_lf_frame(ctx, case, y_only=False):
    y, u, v = case["y"].copy(), case["u"].copy(), case["v"].copy()
    d = _yuv_desc(y, u, v)
    lfm = np.ascontiguousarray(case["lfm"])
    B.check(B.load().svt_hip_lf_frame(ctx, C.byref(d), lfm.ctypes.data_as(C.c_void_p), lfm.shape[1], C.byref(case["thr"]),
                                      case["mi_rows"], case["mi_cols"], int(y_only)))
    return y, u, v


# ---------------------------------------------------------------------------------------------------
# mask construction (L2): mode-info grids
# ---------------------------------------------------------------------------------------------------
_BS_W4 = [1, 1, 2, 2, 2, 4, 4, 4, 8, 8, 8, 16, 16]   # block width / height in 4-sample units by BLOCK_SIZE
_BS_H4 = [1, 2, 1, 2, 4, 2, 4, 8, 4, 8, 16, 8, 16]
_MODE_LF_LUT = [0] * 10 + [1, 1, 0, 1]                 # VPX/vp9_loopfilter.c mode_lf_lut: intra modes, NEAREST, NEAR, ZERO, NEW


def gen_mode_info_grid(seed, mi_rows, mi_cols, mi_stride=None):
    """A consistent random partition of every SB (all 13 block sizes incl. rectangular and sub-8x8) with transform
    size, skip, reference frame, mode and segment per block.  Returns (cells[mi_rows, mi_stride, 6] for the reference
    driver = {sb_type, tx_size, skip, ref_frame0, mode, segment_id}, lvl[8][4][2], mode-info records for the ABI)."""
    rng = np.random.default_rng(seed)
    mi_stride = mi_stride or mi_cols + 3
    cells = np.zeros((mi_rows, mi_stride, 6), np.uint8)
    lvl = rng.choice(np.array([0, 0, 3, 9, 17, 30, 47, 63], np.uint8), size=(8, 4, 2)).astype(np.uint8)

    def put(r, c, h8, w8, bs):
        maxtx = {1: 0, 2: 1, 4: 2, 8: 3, 16: 3}[min(_BS_W4[bs], _BS_H4[bs])]
        inter = int(rng.integers(0, 3) > 0)
        rec = (bs, int(rng.integers(0, maxtx + 1)), int(rng.integers(0, 2)), int(rng.integers(1, 4)) if inter else 0,
               int(rng.integers(10, 14)) if inter else int(rng.integers(0, 10)), int(rng.integers(0, 8)))
        cells[r:min(r + h8, mi_rows), c:min(c + w8, mi_cols)] = rec

    def part(r, c, n):              # n = size in 8x8 units (8, 4, 2, 1)
        if r >= mi_rows or c >= mi_cols:
            return
        sq = {8: 12, 4: 9, 2: 6, 1: 3}[n]
        k = int(rng.integers(0, 4)) if n > 1 else int(rng.integers(0, 4)) + 4
        if n > 1 and k == 0:        # split
            h = n // 2
            for dr in (0, h):
                for dc in (0, h):
                    part(r + dr, c + dc, h)
        elif n > 1 and k == 1:      # horizontal pair (w x h/2): 64x32 / 32x16 / 16x8
            put(r, c, n // 2, n, sq - 1 if n != 2 else 5)
            if r + n // 2 < mi_rows:
                put(r + n // 2, c, n // 2, n, sq - 1 if n != 2 else 5)
        elif n > 1 and k == 2:      # vertical pair: 32x64 / 16x32 / 8x16
            put(r, c, n, n // 2, sq - 2 if n != 2 else 4)
            if c + n // 2 < mi_cols:
                put(r, c + n // 2, n, n // 2, sq - 2 if n != 2 else 4)
        elif n > 1:
            put(r, c, n, n, sq)
        else:                       # 8x8 unit: 8x8, 8x4, 4x8 or 4x4 (one record per unit)
            put(r, c, 1, 1, [3, 2, 1, 0][k - 4])

    for r in range(0, mi_rows, 8):
        for c in range(0, mi_cols, 8):
            part(r, c, 8)
    mi = np.zeros((mi_rows, mi_stride), dtype=B.LF_MODE_INFO_DTYPE)
    mi["sb_type"], mi["tx_size"], mi["skip"] = cells[..., 0], cells[..., 1], cells[..., 2]
    mi["is_inter"] = cells[..., 3] > 0
    lut = np.array(_MODE_LF_LUT, np.uint8)
    # the reference is built without segmentation support: get_filter_level() always reads segment 0 (:242-248)
    mi["filter_level"] = lvl[0, cells[..., 3], lut[cells[..., 4]]]
    return cells, lvl, mi


def _build_masks(fn, mi, mi_rows, mi_cols):
    sb_rows, lfm_stride = (mi_rows + 7) // 8, (mi_cols + 7) // 8 + 1
    lfm = np.zeros((sb_rows, lfm_stride), dtype=B.LF_MASK_DTYPE)
    rc = fn(mi.ctypes.data_as(C.c_void_p), mi.shape[1], mi_rows, mi_cols, lfm.ctypes.data_as(C.c_void_p), lfm_stride)
    assert rc == 0
    return lfm[:, :lfm_stride - 1].copy()


def oracle_lf_build_masks(mi, mi_rows, mi_cols):
    return _build_masks(oracle().svt_oracle_lf_build_masks, mi, mi_rows, mi_cols)


def product_lf_build_masks(mi, mi_rows, mi_cols):
    return _build_masks(B.load().svt_hip_lf_build_masks, mi, mi_rows, mi_cols)


def ref_lf_build_masks(cells, lvl, mi_rows, mi_cols):
    exe = os.path.join(REF_DIR, "ref_lf_frame")
    sb_rows, lfm_stride = (mi_rows + 7) // 8, (mi_cols + 7) // 8 + 1
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        with open(req, "wb") as f:
            f.write(struct.pack("<5i", 0x4D4C5653, mi_rows, mi_cols, cells.shape[1], lfm_stride))
            f.write(np.ascontiguousarray(lvl).tobytes())
            f.write(np.ascontiguousarray(cells).tobytes())
        subprocess.check_call([exe, req, rsp])
        raw = np.fromfile(rsp, dtype=B.LF_MASK_DTYPE)
    return raw.reshape(sb_rows, lfm_stride)[:, :lfm_stride - 1].copy()


def oracle_tq_batch_dist(case):
    """oracle_tq_batch + the per-block coefficient-domain distortion pairs (T3)."""
    recon = np.zeros_like(case["src"])
    q = np.zeros(case["n_coeff"], np.int16)
    dq = np.zeros(case["n_coeff"], np.int16)
    eob = np.zeros(len(case["blocks"]), np.uint16)
    dist = np.zeros((len(case["blocks"]), 2), np.uint64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = oracle().svt_oracle_tq_batch_dist(vp(case["src"]), vp(case["pred"]), vp(recon), vp(case["blocks"]), len(case["blocks"]),
                                           vp(case["qtabs"]), vp(case["iscan"]), vp(q), vp(dq), vp(eob), vp(dist))
    assert rc == 0
    return recon, q, dq, eob, dist


def hip_tq_batch_dist_device(ctx, case):
    """svt_hip_tq_batch_dist_device on device buffers allocated with torch (GPU tests only)."""
    import torch
    lib = B.load()
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
    src, pred, blocks, qt, isc = up(case["src"]), up(case["pred"]), up(case["blocks"]), up(case["qtabs"]), up(case["iscan"])
    recon = torch.zeros(case["src"].size, dtype=torch.uint8, device=dev)
    q = torch.zeros(case["n_coeff"], dtype=torch.int16, device=dev)
    dq = torch.zeros(case["n_coeff"], dtype=torch.int16, device=dev)
    eob = torch.zeros(len(case["blocks"]), dtype=torch.int16, device=dev)
    dist = torch.zeros(2 * len(case["blocks"]), dtype=torch.int64, device=dev)
    cnt = (C.c_int32 * 4)(*[int(v) for v in case["counts"]])
    p = lambda t: C.c_void_p(t.data_ptr())
    B.check(lib.svt_hip_tq_batch_dist_device(ctx, p(src), p(pred), p(recon), p(blocks), cnt, p(qt), p(isc), p(q), p(dq), p(eob), p(dist)))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    return (recon.cpu().numpy().reshape(case["src"].shape), q.cpu().numpy(), dq.cpu().numpy(), eob.cpu().numpy().view(np.uint16),
            dist.cpu().numpy().view(np.uint64).reshape(-1, 2))


def add_rate_info(case, seed, inter_share=0.5):
    """rate inputs for the fused distortion + rate entry: plane type, inter flag and entropy context per block in pad[0]
    (SVT_TQ_RATE_INFO).  Returns the equivalent svt_rate_block records (eob left 0: it is the transform stage's output)."""
    rng = np.random.default_rng(seed)
    nb = len(case["blocks"])
    pt, inter, ctx = rng.integers(0, 2, nb), (rng.random(nb) < inter_share).astype(np.int64), rng.integers(0, 3, nb)
    case["blocks"]["pad"][:, 0] = (ctx | (pt << 2) | (inter << 3)).astype(np.uint8)
    offs, _ = rate_scan_offsets()
    rb = np.zeros(nb, dtype=B.RATE_BLOCK_DTYPE)
    rb["coeff_off"], rb["tx_size"], rb["plane_type"], rb["is_inter"], rb["ctx"] = case["blocks"]["coeff_off"], case["blocks"]["tx_size"], pt, inter, ctx
    rb["scan_off"] = [offs[(int(b["tx_size"]), int(b["tx_type"]) if b["tx_size"] < 3 else 0)] for b in case["blocks"]]
    return rb


def oracle_tq_rd_batch(case, rb):
    """the oracle's two stages chained: transform / quantisation (+ distortion), then coeff_rate_estimate on its output"""
    recon, q, dq, eob, dist = oracle_tq_batch_dist(case)
    rb = rb.copy()
    rb["eob"] = eob
    return recon, q, dq, eob, dist, oracle_rate_batch(dict(qcoeff=q, blocks=rb))


def hip_tq_rd_batch_device(ctx, case, null_recon=False):
    """svt_hip_tq_rd_batch_device on device buffers allocated with torch (GPU tests only).  null_recon: d_recon = NULL (the mode
    decision form: no block reconstructs)."""
    import torch
    lib = B.load()
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
    src, pred, blocks, qt, isc = up(case["src"]), up(case["pred"]), up(case["blocks"]), up(case["qtabs"]), up(case["iscan"])
    rtab, rscan = rate_tables()
    tab, scan = up(np.ascontiguousarray(rtab).reshape(1)), up(rscan)
    recon = torch.zeros(case["src"].size, dtype=torch.uint8, device=dev)
    q = torch.zeros(case["n_coeff"], dtype=torch.int16, device=dev)
    dq = torch.zeros(case["n_coeff"], dtype=torch.int16, device=dev)
    eob = torch.zeros(len(case["blocks"]), dtype=torch.int16, device=dev)
    dist = torch.zeros(2 * len(case["blocks"]), dtype=torch.int64, device=dev)
    bits = torch.zeros(len(case["blocks"]), dtype=torch.int32, device=dev)
    cnt = (C.c_int32 * 4)(*[int(v) for v in case["counts"]])
    p = lambda t: C.c_void_p(t.data_ptr())
    B.check(lib.svt_hip_tq_rd_batch_device(ctx, p(src), p(pred), None if null_recon else p(recon), p(blocks), cnt, p(qt), p(isc), p(q), p(dq),
                                           p(eob), p(dist), p(tab), p(scan), p(bits)))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    return (recon.cpu().numpy().reshape(case["src"].shape), q.cpu().numpy(), dq.cpu().numpy(), eob.cpu().numpy().view(np.uint16),
            dist.cpu().numpy().view(np.uint64).reshape(-1, 2), bits.cpu().numpy())


# ---------------------------------------------------------------------------------------------------
# M12 side outputs
# ---------------------------------------------------------------------------------------------------
def oracle_me_zz_sad(cur, prev, input_resolution):
    """cur, prev: PaPic of the current and the previous picture.  Returns (zz_sad[n_sb], non_moving_index[n_sb])."""
    dc, dp = cur.desc(), prev.desc()
    n = n_sb(prev.luma.shape[1], prev.luma.shape[0])
    zz, nmi = np.zeros(n, np.uint32), np.zeros(n, np.uint8)
    rc = oracle().svt_oracle_me_zz_sad(C.byref(dc.sixteenth), C.byref(dp.full), input_resolution, zz.ctypes.data_as(C.c_void_p),
                                       nmi.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return zz, nmi


def hip_me_zz_sad(ctx, cur, prev, input_resolution):
    import torch
    lib = B.load()
    dev = torch.device("cuda", 0)
    (c16, pad16), (pf, padf) = cur.planes()[2], prev.planes()[0]
    t16, tf = torch.from_numpy(np.ascontiguousarray(c16)).to(dev), torch.from_numpy(np.ascontiguousarray(pf)).to(dev)
    n = n_sb(prev.luma.shape[1], prev.luma.shape[0])
    zz, nmi = torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
    a, b = B.plane_desc(c16, pad16, pad16, ptr=t16.data_ptr()), B.plane_desc(pf, padf, padf, ptr=tf.data_ptr())
    B.check(lib.svt_hip_me_zz_sad_device(ctx, C.byref(a), C.byref(b), input_resolution, C.c_void_p(zz.data_ptr()), C.c_void_p(nmi.data_ptr())))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    return zz.cpu().numpy().view(np.uint32), nmi.cpu().numpy()


# ---- inter prediction (8-tap motion compensation) -------------------------------------------------------------
def make_mc_case(seed, width=192, height=128, pad=80, mv_range=48, use_subpel=1, rect=True, intra_share=0.1):
    """Two padded reference pictures, a mode-info grid cut into VP9 partitions (64..8, optionally rectangular) with a
    mix of single-list / compound blocks, zero / full-sample / sub-sample / far-out-of-picture MVs, some intra units."""
    rng = np.random.default_rng(seed)
    mi_rows, mi_cols = height // 8, width // 8
    refs = []
    for l in range(2):
        y, u, v = gen_yuv(width, height, seed * 7 + l)
        v = (255 - u[::-1, ::-1] // 2 - v // 4).astype(np.uint8)   # gen_yuv's V plane is nearly flat: give it texture
        y = np.ascontiguousarray(np.pad(y, pad, mode="edge"))
        u = np.ascontiguousarray(np.pad(u, pad // 2, mode="edge"))
        v = np.ascontiguousarray(np.pad(v, pad // 2, mode="edge"))
        refs.append((y, u, v))
    mi = np.zeros((mi_rows, mi_cols), dtype=B.MC_MODE_INFO_DTYPE)
    mi["ref_list"] = -1

    def fill(r, c, h8, w8):
        kind = rng.random()
        cell = np.zeros((), dtype=B.MC_MODE_INFO_DTYPE)
        cell["bw8"], cell["bh8"] = w8, h8
        if kind < intra_share:
            cell["ref_list"] = (-1, -1)
        else:
            comp = rng.random() < 0.4
            l0 = int(rng.integers(0, 2))
            cell["ref_list"] = (l0, (1 - l0 if rng.random() < 0.8 else l0) if comp else -1)
            for k in range(2):
                t = rng.random()
                if t < 0.15: mv = (0, 0)
                elif t < 0.35: mv = tuple(int(x) * 8 for x in rng.integers(-mv_range // 8, mv_range // 8 + 1, 2))
                elif t < 0.9: mv = tuple(int(x) for x in rng.integers(-mv_range * 8, mv_range * 8 + 1, 2))
                else: mv = tuple(int(x) for x in rng.integers(-4000, 4001, 2))  # far outside: clamp_mv_to_umv_border_sb
                cell["mv_row"][k], cell["mv_col"][k] = mv
        mi[r:r + h8, c:c + w8] = cell

    def split(r, c, n8):
        if r >= mi_rows or c >= mi_cols:
            return
        fits = r + n8 <= mi_rows and c + n8 <= mi_cols
        t = rng.random()
        if n8 > 1 and (not fits or t < 0.55):
            h = n8 // 2
            for dr, dc in ((0, 0), (0, h), (h, 0), (h, h)):
                split(r + dr, c + dc, h)
        elif rect and n8 > 1 and t < 0.7:
            fill(r, c, n8 // 2, n8); fill(r + n8 // 2, c, n8 // 2, n8)       # PARTITION_HORZ
        elif rect and n8 > 1 and t < 0.85:
            fill(r, c, n8, n8 // 2); fill(r, c + n8 // 2, n8, n8 // 2)       # PARTITION_VERT
        else:
            fill(r, c, n8, n8)

    for r in range(0, mi_rows, 8):
        for c in range(0, mi_cols, 8):
            split(r, c, 8)
    return dict(mi=mi, mi_rows=mi_rows, mi_cols=mi_cols, refs=refs, pad=pad, use_subpel=use_subpel, width=width, height=height)


def _mc_host_refs(case):
    arr = (B.McHostRef * 2)()
    for l, (y, u, v) in enumerate(case["refs"]):
        arr[l].y, arr[l].u, arr[l].v = y.ctypes.data, u.ctypes.data, v.ctypes.data
        arr[l].y_stride, arr[l].uv_stride = y.shape[1], u.shape[1]
        arr[l].org_x = arr[l].org_y = case["pad"]
    return arr


def _mc_out(case, fill=0x5A):
    W, H = case["width"], case["height"]
    return [np.full((H, W), fill, np.uint8), np.full((H // 2, W // 2), fill, np.uint8), np.full((H // 2, W // 2), fill, np.uint8)]


def oracle_mc_frame(case):
    lib = oracle()
    mi = np.ascontiguousarray(case["mi"])
    out = _mc_out(case)
    rc = lib.svt_oracle_inter_pred_frame(C.c_void_p(mi.ctypes.data), case["mi_cols"], case["mi_rows"], case["mi_cols"], _mc_host_refs(case),
                                         case["use_subpel"], *[C.c_void_p(o.ctypes.data) for o in out])
    assert rc == 0
    return out


def hip_mc_frame(ctx, case):
    lib = B.load()
    mi = np.ascontiguousarray(case["mi"])
    out = _mc_out(case)
    B.check(lib.svt_hip_inter_pred_frame(ctx, C.c_void_p(mi.ctypes.data), case["mi_cols"], case["mi_rows"], case["mi_cols"], _mc_host_refs(case),
                                         case["use_subpel"], *[C.c_void_p(o.ctypes.data) for o in out]))
    return out


def _run_timed(exe, req, td, n, env=None):
    """runs `exe req rsp_k` n times at once (one process each) and returns the n timing doubles the harnesses append to their responses"""
    from concurrent.futures import ThreadPoolExecutor

    def one(k):
        rsp = os.path.join(td, f"rsp{k}.bin")
        subprocess.check_call([exe, req, rsp], env=env)
        with open(rsp, "rb") as f:
            f.seek(-8, 2)
            return struct.unpack("<d", f.read(8))[0]
    with ThreadPoolExecutor(n) as ex:
        return list(ex.map(one, range(n)))


def ref_mc_frame(case, asm_type=0, timing=0):
    """the reference's own inter_prediction() for every block (oracle/_ref/ref_mc_frame); units it does not predict are 0"""
    exe = os.path.join(REF_DIR, "ref_mc_frame")
    mi = np.ascontiguousarray(case["mi"])
    W, H = case["width"], case["height"]
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        with open(req, "wb") as f:
            f.write(struct.pack("<6i", 0x434D5653, case["mi_rows"], case["mi_cols"], case["mi_cols"], case["use_subpel"], asm_type))
            for (y, u, v) in case["refs"]:
                f.write(struct.pack("<6i", y.shape[1], u.shape[1], case["pad"], case["pad"], y.shape[0], u.shape[0]))
                f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())
            f.write(mi.tobytes())
        if timing:
            return _run_timed(exe, req, td, timing)
        subprocess.check_call([exe, req, rsp])
        raw = open(rsp, "rb").read()
    return [np.frombuffer(raw, np.uint8, W * H).reshape(H, W).copy(),
            np.frombuffer(raw, np.uint8, W * H // 4, W * H).reshape(H // 2, W // 2).copy(),
            np.frombuffer(raw, np.uint8, W * H // 4, W * H + W * H // 4).reshape(H // 2, W // 2).copy()]


def mc_inter_masks(case):
    """boolean masks (luma, chroma) of the samples that belong to inter blocks"""
    inter = case["mi"]["ref_list"][:, :, 0] >= 0
    return np.kron(inter, np.ones((8, 8), bool)), np.kron(inter, np.ones((4, 4), bool))


# ---- coefficient rate estimation ---------------------------------------------------------------------------------
RATE_GOLD = os.path.join(GOLDEN_DIR, "rate_reference.npz")


def _rate_request(case):
    """request file of oracle/_ref/ref_rate_blocks"""
    out = [struct.pack("<2i", 0x54525653, len(case["blocks"]))]
    for b, tt in zip(case["blocks"], case["tx_type"]):
        n = 16 << (2 * int(b["tx_size"]))
        plane = 0 if b["plane_type"] == 0 else 1
        out.append(struct.pack("<6i", int(b["tx_size"]), plane, int(b["is_inter"]), int(tt), int(b["ctx"]), int(b["eob"])))
        out.append(case["qcoeff"][int(b["coeff_off"]):int(b["coeff_off"]) + n].astype("<i2").tobytes())
    return b"".join(out)


def ref_rate_run(case):
    """the reference's coeff_rate_estimate() for every block + the tables it used (tables, scan array, offsets)"""
    exe = os.path.join(REF_DIR, "ref_rate_blocks")
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        open(req, "wb").write(_rate_request(case))
        subprocess.check_call([exe, req, rsp])
        raw = open(rsp, "rb").read()
    nb = len(case["blocks"])
    bits = np.frombuffer(raw, "<i4", nb).copy()
    pos = 4 * nb
    t = np.zeros((), dtype=B.RATE_TABLES_DTYPE)
    t["token_costs"] = np.frombuffer(raw, "<u4", 13824, pos).reshape(4, 2, 2, 6, 2, 6, 12); pos += 4 * 13824
    t["value_cost"] = np.frombuffer(raw, "<i4", 133, pos); pos += 4 * 133
    t["cat6_low_cost"] = np.frombuffer(raw, "<u2", 256, pos); pos += 512
    t["cat6_high_cost"] = np.frombuffer(raw, "<u2", 64, pos); pos += 128
    scan = np.frombuffer(raw, "<i2", (len(raw) - pos) // 2, pos).copy()
    return bits, t, scan


def rate_scan_offsets():
    offs, pos = {}, 0
    for ts in range(4):
        n = 16 << (2 * ts)
        for tt in range(4):
            offs[(ts, tt)] = pos
            pos += n + 2 * (n + 1)
    return offs, pos


_rate_tab_cache = None


def rate_tables():
    """(svt_rate_tables, scan array) as committed test data (tests/golden/rate_reference.npz, written by gen_golden.py from
    the reference's own tables)"""
    global _rate_tab_cache
    if _rate_tab_cache is None:
        g = np.load(RATE_GOLD)
        t = np.zeros((), dtype=B.RATE_TABLES_DTYPE)
        for k in ("token_costs", "value_cost", "cat6_low_cost", "cat6_high_cost"):
            t[k] = g[k]
        _rate_tab_cache = (t, g["scan"].astype(np.int16))
    return _rate_tab_cache


def make_rate_case(seed, width=256, height=128, scan=None, extreme=False):
    """quantised transform blocks out of the TQ oracle (realistic sparsity and eob spread) + per-block plane type, inter flag,
    entropy context; tx_type of inter / chroma blocks is DCT_DCT as get_tx_type() has it.  A few blocks get huge levels
    (CAT6 tokens) and dense 32x32 content."""
    rng = np.random.default_rng(seed)
    tq = make_tq_case(seed, width=width, height=height, qsteps=((8, 9), (16, 20), (40, 48)) if not extreme else ((4, 4), (4, 4), (8, 8)), extreme=extreme)
    _, q, _, _ = oracle_tq_batch(tq)
    q = q.copy()
    offs, total = rate_scan_offsets()
    if scan is None:
        scan = rate_tables()[1]
    assert scan.size == total
    nb = len(tq["blocks"])
    blocks = np.zeros(nb, dtype=B.RATE_BLOCK_DTYPE)
    tx_type = np.zeros(nb, np.int32)
    for i, b in enumerate(tq["blocks"]):
        ts = int(b["tx_size"])
        n = 16 << (2 * ts)
        off = int(b["coeff_off"])
        pt, inter = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        tt = int(b["tx_type"]) if (pt == 0 and not inter and ts < 3) else 0
        if rng.random() < 0.02:
            q[off:off + n][rng.integers(0, n, 3)] = rng.integers(-3000, 3000, 3)   # CAT6 / large tokens
        so = offs[(ts, tt)]
        sc = scan[so:so + n]
        nz = np.nonzero(q[off:off + n][sc])[0]
        blocks[i] = (off, so, (int(nz[-1]) + 1) if len(nz) else 0, ts, pt, inter, int(rng.integers(0, 3)), (0, 0))
        tx_type[i] = tt
    return dict(qcoeff=q, blocks=blocks, tx_type=tx_type)


def oracle_rate_batch(case, tables=None, scan=None):
    t, s = (tables, scan) if tables is not None else rate_tables()
    bits = np.zeros(len(case["blocks"]), np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    tb = np.ascontiguousarray(t).reshape(1)
    rc = oracle().svt_oracle_coeff_rate_batch(vp(case["qcoeff"]), vp(case["blocks"]), len(case["blocks"]), vp(tb), vp(s), vp(bits))
    assert rc == 0
    return bits


def hip_rate_batch(ctx, case):
    t, s = rate_tables()
    bits = np.zeros(len(case["blocks"]), np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    tb = np.ascontiguousarray(t).reshape(1)
    B.check(B.load().svt_hip_coeff_rate_batch(ctx, vp(case["qcoeff"]), C.c_size_t(case["qcoeff"].size), vp(case["blocks"]), len(case["blocks"]),
                                              vp(tb), vp(s), C.c_size_t(s.size), vp(bits)))
    return bits


# ---- M12, rest: stationary-edge flags and rate-control SAD-interval histograms -------------------------------------
def make_sb_stats_case(seed, width, height, input_resolution, temporal_layer, slice_type, run_part2, rate_control_mode=1):
    rng = np.random.default_rng(seed)
    n = n_sb(width, height)
    res = np.zeros((n, 85), dtype=B.ME_RESULT_DTYPE)
    res["x_mv_l0"][:, 0], res["y_mv_l0"][:, 0] = rng.integers(-40, 41, n), rng.integers(-40, 41, n)
    res["dist0"][:, 0] = rng.integers(0, 4, n) * 6000 + rng.integers(0, 9000, n)       # around 64 * 64 * {2, 5}
    var = rng.integers(0, 3000, (n, 85)).astype(np.uint16)
    # wide spreads up to the largest variance 8-bit samples can have (127.5^2 = 16256): beyond that the reference's int32
    # products of differences overflow, which is undefined behaviour in its C (the compiled result depends on the compiler)
    var[::3, 1:5] = rng.integers(0, 16257, (len(var[::3]), 4))
    var[::7, 1:5] = np.array([0, 0, 0, 16256], np.uint16)
    var[1::7, 1:5] = rng.integers(1000, 1040, (len(var[1::7]), 4))                     # near-equal: both checks off
    var[:, 0] = rng.integers(0, 65536, n)
    rcme = rng.integers(0, 1 << 20, n).astype(np.uint32)
    p = B.MeSbStatsParams(width, height, input_resolution, temporal_layer, slice_type, run_part2, rate_control_mode)
    return dict(p=p, res=res, var=var, rcme=rcme, n=n)


def oracle_me_sb_stats(case):
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    out = np.zeros(case["n"], dtype=B.ME_SB_STATS_DTYPE)
    hist, full = np.zeros(256, np.uint32), np.zeros(1, np.uint32)
    rc = oracle().svt_oracle_me_sb_stats(C.byref(case["p"]), vp(case["res"]), vp(case["var"]), vp(case["rcme"]), vp(out), vp(hist), vp(full))
    assert rc == 0
    return out, hist, int(full[0])


def hip_me_sb_stats(ctx, case):
    import torch
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
    res, var, rcme = up(case["res"]), up(case["var"]), up(case["rcme"])
    out = torch.zeros(case["n"] * 8, dtype=torch.uint8, device=dev)
    hist, full = torch.zeros(256, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    lib = B.load()
    B.check(lib.svt_hip_me_sb_stats_device(ctx, C.byref(case["p"]), p(res), p(var), p(rcme), p(out), p(hist), p(full)))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    return out.cpu().numpy().view(B.ME_SB_STATS_DTYPE), hist.cpu().numpy().view(np.uint32), int(full.cpu().numpy()[0])


def ref_me_stationary_edge(case):
    """the reference's own stationary_edge_over_update_over_time_sb_part1 / _part2 and eb_vp9_sb_params_init
    (oracle/_ref/ref_me_side, request 'SVMT'): uint8 [n_sb][6] = check1, pm_check1, check2, low_dist_logo, potential_logo_sb, complete"""
    exe = os.path.join(REF_DIR, "ref_me_side")
    p = case["p"]
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        with open(req, "wb") as f:
            f.write(struct.pack("<7i", 0x544D5653, p.pic_width, p.pic_height, p.input_resolution, p.temporal_layer_index, p.slice_type, p.run_part2))
            for i in range(case["n"]):
                r = case["res"][i, 0]
                f.write(struct.pack("<hhI4H", int(r["x_mv_l0"]), int(r["y_mv_l0"]), int(r["dist0"]), *[int(v) for v in case["var"][i, 1:5]]))
        subprocess.check_call([exe, req, rsp])
        return np.frombuffer(open(rsp, "rb").read(), np.uint8).reshape(case["n"], 6).copy()


SB_STATS_CASES = ((1, 640, 360, 0, 0, 2, 1), (2, 1280, 720, 1, 1, 0, 1), (3, 1920, 1080, 2, 3, 0, 1), (4, 3840, 2160, 3, 2, 0, 1), (5, 3840, 2160, 3, 0, 1, 0),
                  (6, 328, 200, 0, 4, 0, 1), (7, 1920, 1088, 2, 0, 2, 0), (8, 832, 480, 0, 1, 1, 1))


def ref_me_side(cur, prev, input_resolution, cur_mean, cur_var, ref_mean, ref_var, is_i_slice, is_used_as_reference):
    """the reference's compute_zz_sad + eb_vp9_derive_similar_collocated_flag (oracle/_ref/ref_me_side).
    Returns (non_moving_index, similar, similar_all_layers)."""
    exe = os.path.join(REF_DIR, "ref_me_side")
    h, w = prev.luma.shape
    n = n_sb(w, h)
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        with open(req, "wb") as f:
            f.write(struct.pack("<6i", 0x534D5653, w, h, input_resolution, int(is_i_slice), int(is_used_as_reference)))
            for plane, org in ((cur.sixteenth, 16), (prev.full, 68)):
                f.write(struct.pack("<4i", plane.shape[1], org, org, plane.shape[0]))
                f.write(plane.tobytes())
            f.write(struct.pack("<i", n))
            f.write(cur_mean.astype(np.uint8).tobytes()); f.write(cur_var.astype("<u2").tobytes())
            f.write(ref_mean.astype(np.uint8).tobytes()); f.write(ref_var.astype("<u2").tobytes())
        subprocess.check_call([exe, req, rsp])
        raw = np.frombuffer(open(rsp, "rb").read(), np.uint8)
    return raw[:n].copy(), raw[n:2 * n].copy(), raw[2 * n:3 * n].copy()


def ref_ivf_headers(width, height, frame_rate_q16, numerator, denominator, frames):
    """bytes written by the REFERENCE application's write_ivf_stream_header followed by one write_ivf_frame_header per
    (byte_count, pts) (oracle/_ref/ref_ivf_headers = App/EbAppProcessCmd.c compiled as it lies)"""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "h.bin")
        args = [os.path.join(REF_DIR, "ref_ivf_headers"), out, str(width), str(height), str(frame_rate_q16), str(numerator), str(denominator)]
        for n, pts in frames:
            args += [str(n), str(pts)]
        subprocess.check_call(args)
        return open(out, "rb").read()


# ---------------------------------------------------------------------------------------------------
# deblocked reconstruction -> padded reference picture (pad_ref_and_set_flags, Codec/EbEncDecProcess.c:4822-4851)
# ---------------------------------------------------------------------------------------------------
REFPAD_GOLDEN_CASES = ((1, 72, 40, 80, 80, 0), (2, 200, 136, 80, 80, 8), (3, 64, 64, 16, 32, 0), (4, 136, 72, 70, 10, 3))


def make_refpad_case(seed, width, height, pad_x=80, pad_y=80, slack=0):
    """three padded buffers whose borders hold junk (what a recycled reference buffer holds before the padding)"""
    rng = np.random.default_rng(seed)
    bufs = []
    for sh in (0, 1, 1):
        w, h, px, py = width >> sh, height >> sh, pad_x >> sh, pad_y >> sh
        bufs.append(rng.integers(0, 256, (h + 2 * py, w + 2 * px + slack), dtype=np.uint8))
    return dict(bufs=bufs, width=width, height=height, pad_x=pad_x, pad_y=pad_y, slack=slack)


def _refpad_desc(case, ptrs):
    d = B.YuvPlanes()
    px, py = case["pad_x"], case["pad_y"]
    sy, sc = case["bufs"][0].shape[1], case["bufs"][1].shape[1]
    d.y = ptrs[0] + py * sy + px
    d.u, d.v = (p + (py >> 1) * sc + (px >> 1) for p in ptrs[1:])
    d.y_stride, d.uv_stride, d.width, d.height = sy, sc, case["width"], case["height"]
    return d


def refpad_valid(case, bufs):
    """the part of each buffer the padding defines: every row, the first width + 2 pad bytes (not the slack behind them)"""
    return [b[:, :b.shape[1] - case["slack"]] for b in bufs]


def oracle_ref_pad(case):
    out = [b.copy() for b in case["bufs"]]
    d = _refpad_desc(case, [o.ctypes.data for o in out])
    assert oracle().svt_oracle_ref_pad(C.byref(d), case["pad_x"], case["pad_y"]) == 0
    return out


def ref_ref_pad(case):
    """the reference's own pad_ref_and_set_flags (oracle/_ref/ref_refpad)"""
    with tempfile.TemporaryDirectory() as td:
        rq, rs = os.path.join(td, "rq"), os.path.join(td, "rs")
        with open(rq, "wb") as f:
            f.write(struct.pack("<7i", 0x50525653, case["width"], case["height"], case["pad_x"], case["pad_y"], case["bufs"][0].shape[1], case["bufs"][1].shape[1]))
            for b in case["bufs"]:
                f.write(np.ascontiguousarray(b).tobytes())
        subprocess.check_call([os.path.join(REF_DIR, "ref_refpad"), rq, rs])
        raw = np.fromfile(rs, np.uint8)
    out, pos = [], 0
    for b in case["bufs"]:
        out.append(raw[pos:pos + b.size].reshape(b.shape).copy())
        pos += b.size
    return out


def ref_encdec_flags():
    """the reference's own eb_vp9_signal_derivation_enc_dec_kernel_{sq,oq,vmaf} (oracle/_ref/ref_refpad, request 'SVFL'):
    [tune 0..2][enc_mode 0..12][temporal layer 0..4][is_used_as_reference 0..1][limit_intra, allow_enc_dec_mismatch]"""
    with tempfile.TemporaryDirectory() as td:
        rq, rs = os.path.join(td, "rq"), os.path.join(td, "rs")
        with open(rq, "wb") as f:
            f.write(struct.pack("<i", 0x4C465653))
        subprocess.check_call([os.path.join(REF_DIR, "ref_refpad"), rq, rs])
        return np.fromfile(rs, np.uint8).reshape(3, 13, 5, 2, 2)


def hip_ref_pad_batch(ctx, cases):
    """several pictures (of different sizes) through one svt_hip_ref_pad_batch_device call; pad_x / pad_y of the first case"""
    import torch
    lib = B.load()
    dev = [[torch.from_numpy(np.ascontiguousarray(b)).cuda() for b in c["bufs"]] for c in cases]
    descs = (B.YuvPlanes * len(cases))(*[_refpad_desc(c, [t.data_ptr() for t in ts]) for c, ts in zip(cases, dev)])
    torch.cuda.synchronize()
    B.check(lib.svt_hip_ref_pad_batch_device(ctx, len(cases), descs, cases[0]["pad_x"], cases[0]["pad_y"]))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    return [[t.cpu().numpy() for t in ts] for ts in dev]


# ---------------------------------------------------------------------------------------------------
# eb_vp9_combined_averaging_ssd (Codec/EbMotionEstimation.c:1708-1725): the quarter-pel metric of the SSD fractional search
# ---------------------------------------------------------------------------------------------------
def make_avg_ssd_jobs(seed, n=40):
    """(src, ref1, ref2) blocks of the PU sizes, strides >= width; includes the extremes (0 / 255 planes: the largest sums)"""
    rng = np.random.default_rng(seed)
    jobs = []
    for i in range(n):
        w = int(rng.choice([8, 16, 32, 64]))
        h = w if i % 5 else w // 2
        strides = [w + int(rng.integers(0, 3)) * 8 for _ in range(3)]
        if i % 7 == 0:
            vals = [(0, 255, 255), (255, 0, 0), (255, 0, 1)][(i // 7) % 3]
            blocks = [np.full((h, st), v, np.uint8) for st, v in zip(strides, vals)]
        else:
            blocks = [rng.integers(0, 256, (h, st), dtype=np.uint8) for st in strides]
        jobs.append((w, h, blocks))
    return jobs


def oracle_avg_ssd_jobs(jobs):
    o = oracle()
    o.oracle_avg_ssd.restype = C.c_uint32
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    return np.array([o.oracle_avg_ssd(vp(b[0]), b[0].shape[1], vp(b[1]), b[1].shape[1], vp(b[2]), b[2].shape[1], h, w) for w, h, b in jobs], np.uint32)


def ref_avg_ssd_jobs(jobs):
    """the reference's own eb_vp9_combined_averaging_ssd (request 'SVAS' of oracle/_ref/ref_me_sb)"""
    with tempfile.TemporaryDirectory() as td:
        rq, rs = os.path.join(td, "rq"), os.path.join(td, "rs")
        with open(rq, "wb") as f:
            f.write(struct.pack("<2i", 0x53415653, len(jobs)))
            for w, h, b in jobs:
                f.write(struct.pack("<5i", w, h, b[0].shape[1], b[1].shape[1], b[2].shape[1]))
                for a in b:
                    f.write(np.ascontiguousarray(a).tobytes())
        subprocess.check_call([os.path.join(REF_DIR, "ref_me_sb"), rq, rs])
        return np.fromfile(rs, np.uint32)


# ---------------------------------------------------------------------------------------------------
# mini-GOP window split of the picture-decision kernel (Codec/EbPictureDecisionProcess.c:367-476, 1662-1680)
# ---------------------------------------------------------------------------------------------------
class MinigopPart(C.Structure):
    _fields_ = [("start", C.c_int32), ("length", C.c_int32), ("hierarchical_levels", C.c_int32), ("random_access", C.c_int32)]


def ref_minigop_split():
    """the reference's own split of pre-assignment buffers of 2..16 pictures: rows (n, start, length, levels), padded with -1"""
    out = subprocess.check_output([os.path.join(REF_DIR, "ref_pd_split")]).decode()
    rows = np.full((15, 1 + 3 * 4), -1, np.int32)
    for k, line in enumerate(out.strip().splitlines()):
        v = [int(x) for x in line.split()]
        rows[k, 0] = v[0]
        rows[k, 1:1 + 3 * v[1]] = v[2:]
    return rows


def product_minigop_split(n, levels=4, cut_by_intra=0):
    parts = (MinigopPart * 4)()
    k = B.load().svt_hip_minigop_split(n, levels, cut_by_intra, parts)
    assert k >= 1
    return [(p.start, p.length, p.hierarchical_levels, p.random_access) for p in parts[:k]]


# ---- the reference's ME kernel process itself + this repository's binding at its call site (oracle/_ref/ref_me_process) ----
def ref_me_process(cur, ref0, ref1, enc_mode, tune, temporal_layer, p_slice=0, used=1, rate_control_mode=1, same_ref_poc=0, segments=(2, 2), stats=None,
                   run_binding=False, device=0):
    """Runs eb_vp9_motion_estimation_kernel (Codec/EbMotionEstimationProcess.c:875-1290) as a thread behind the reference's own FIFOs on one
    picture and, optionally, integration/me_process_binding.h (-> svt_hip_me_picture) on the same control sets.  stats: dict of per-SB arrays
    cur_mean (u8), var (u16 [n_sb][5]: 64x64, four 32x32), ref_mean (u8), ref_var (u16).  Returns a dict."""
    exe = os.path.join(REF_DIR, "ref_me_process")
    h, w = cur.luma.shape
    nsb = n_sb(w, h)
    if stats is None:
        stats = dict(cur_mean=np.zeros(nsb, np.uint8), var=np.zeros((nsb, 5), np.uint16), ref_mean=np.zeros(nsb, np.uint8), ref_var=np.zeros(nsb, np.uint16))
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        with open(req, "wb") as f:
            f.write(struct.pack("<15i", 0x504D5653, w, h, enc_mode, tune, temporal_layer, p_slice, used, rate_control_mode, same_ref_poc, segments[0], segments[1],
                                int(run_binding), device, 0))
            for pic in (cur, ref0, ref1 if ref1 is not None else ref0):
                for arr, pad in pic.planes():
                    hh, ww = arr.shape
                    f.write(struct.pack("<6i", ww, pad, pad, ww - 2 * pad, hh - 2 * pad, arr.size))
                    f.write(arr.tobytes())
            for i in range(nsb):
                f.write(struct.pack("<B5HBH", int(stats["cur_mean"][i]), *[int(v) for v in stats["var"][i]], int(stats["ref_mean"][i]), int(stats["ref_var"][i])))
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "svt-vp9_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
        subprocess.check_call([exe, req, rsp], env=env)
        raw = open(rsp, "rb").read()
    n, n_sad, n_intra = struct.unpack_from("<3i", raw, 0)
    assert n == nsb
    o = 12
    out = {}

    def take(dtype, count, shape=None):
        nonlocal o
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=o).copy()
        o += a.nbytes
        return a.reshape(shape) if shape else a
    out["res"] = take(B.ME_RESULT_DTYPE, nsb * 85, (nsb, 85))
    out["rcme"] = take(np.uint32, nsb)
    out["inter_idx"], out["intra_idx"] = take(np.uint32, nsb), take(np.uint32, nsb)
    out["me_hist"], out["ois_hist"] = take(np.uint16, n_sad), take(np.uint16, n_intra)
    out["full_sb_count"] = int(take(np.uint32, 1)[0])
    out["similar"], out["similar_all"], out["check1"], out["pm_check1"] = (take(np.uint8, nsb) for _ in range(4))
    out["binding_params"] = B.MeParams.from_buffer_copy(raw[o:o + C.sizeof(B.MeParams)])
    o += C.sizeof(B.MeParams)
    out["binding_rc"] = struct.unpack_from("<i", raw, o)[0]
    o += 4
    if run_binding:
        out["binding_res"] = take(B.ME_RESULT_DTYPE, nsb * 85, (nsb, 85))
        out["binding_rcme"] = take(np.uint32, nsb)
    return out


def ref_lf_call_site(y, u, v, cells, filter_level, sharpness=0, y_only=False, run_binding=False, device=0, timing=0):
    """The deblocking call site of the reference's encode pass (Codec/EbEncDecProcess.c:5676-5686) on a real VP9_COMMON / MACROBLOCKD
    (oracle/_ref/ref_lf_binding): eb_vp9_build_mask_frame + eb_vp9_loop_filter_frame, and -- run_binding -- the same with
    integration/loop_filter_binding.h in place of the second call.  y / u / v: picture planes (W x H, W/2 x H/2); cells: [mi_rows][mi_cols][6].
    Returns (reference planes, binding planes or None, binding rc, LOOP_FILTER_MASK array [sb_rows][sb_cols])."""
    exe = os.path.join(REF_DIR, "ref_lf_binding")
    H, W = y.shape
    # the reference filters whole 8-sample groups and relies on the recon buffer's padding where a chroma block is only 4 samples wide / high
    yp = np.ascontiguousarray(np.pad(y, ((0, 32), (0, 32)), mode="edge"))
    up = np.ascontiguousarray(np.pad(u, ((0, 16), (0, 16)), mode="edge"))
    vp_ = np.ascontiguousarray(np.pad(v, ((0, 16), (0, 16)), mode="edge"))
    mi_rows, mi_cols = H // 8, W // 8
    sb_rows, sb_cols = (mi_rows + 7) // 8, (mi_cols + 7) // 8
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        with open(req, "wb") as f:
            f.write(struct.pack("<13i", 0x424C5653, W, H, yp.shape[1], up.shape[1], yp.shape[0], up.shape[0], filter_level, sharpness, int(y_only), int(run_binding), device, 0))
            f.write(np.ascontiguousarray(cells[:mi_rows, :mi_cols]).tobytes())
            f.write(yp.tobytes()); f.write(up.tobytes()); f.write(vp_.tobytes())
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "svt-vp9_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
        if timing:
            return _run_timed(exe, req, td, timing, env)
        subprocess.check_call([exe, req, rsp], env=env)
        raw = open(rsp, "rb").read()
    ys, us = yp.size, up.size

    def planes(o):
        return (np.frombuffer(raw, np.uint8, ys, o).reshape(yp.shape)[:H, :W].copy(), np.frombuffer(raw, np.uint8, us, o + ys).reshape(up.shape)[:H // 2, :W // 2].copy(),
                np.frombuffer(raw, np.uint8, us, o + ys + us).reshape(vp_.shape)[:H // 2, :W // 2].copy())
    ref = planes(0)
    o = ys + 2 * us
    brc = struct.unpack_from("<i", raw, o)[0]
    o += 4
    bind = None
    if run_binding:
        bind = planes(o)
        o += ys + 2 * us
    lfm = np.frombuffer(raw, dtype=B.LF_MASK_DTYPE, count=sb_rows * sb_cols, offset=o).reshape(sb_rows, sb_cols).copy()
    return ref, bind, brc, lfm


def ref_coding_loop_call_sites(src, pred, lf_mi, q_index, run_binding=False, device=0, timing=0):
    """The transform call sites of the reference's encode pass (Codec/EbEncDecProcess.c:3830, 3890, 3940) on an inter picture
    (oracle/_ref/ref_tq_binding): the reference's own perform_coding_loop per transform block and -- run_binding -- integration/
    coding_loop_binding.h (append per call site, one svt_hip_tq_batch).  src / pred: (Y, U, V) tight planes; lf_mi: [mi_rows][mi_stride]
    svt_lf_mode_info (square inter blocks).  Returns (reference, binding or None, binding rc): each side is a dict with `blocks` (records
    plane, tx_size, x, y, eob), `q`, `dq` (per-block coefficient runs concatenated in call order) and `rec` (Y, U, V)."""
    exe = os.path.join(REF_DIR, "ref_tq_binding")
    H, W = src[0].shape
    rec_dt = np.dtype([("plane", "u1"), ("tx_size", "u1"), ("x", "<u2"), ("y", "<u2"), ("eob", "<u2")])
    with tempfile.TemporaryDirectory() as td:
        req, rsp = os.path.join(td, "req.bin"), os.path.join(td, "rsp.bin")
        with open(req, "wb") as f:
            f.write(struct.pack("<7i", 0x42545653, W, H, lf_mi.shape[1], q_index, -1 if timing else int(run_binding), device))
            for planes in (src, pred):
                for p_ in planes:
                    f.write(np.ascontiguousarray(p_).tobytes())
            f.write(np.ascontiguousarray(lf_mi).tobytes())
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "svt-vp9_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
        if timing:
            return _run_timed(exe, req, td, timing, env)
        subprocess.check_call([exe, req, rsp], env=env)
        raw = open(rsp, "rb").read()
    nb = struct.unpack_from("<i", raw, 0)[0]
    o = 4
    ny, nc = W * H, W * H // 4

    def side():
        nonlocal o
        blocks = np.frombuffer(raw, rec_dt, nb, o).copy()
        o += nb * rec_dt.itemsize
        ncoef = int((16 << (2 * blocks["tx_size"].astype(np.int64))).sum())
        q = np.frombuffer(raw, np.int16, ncoef, o).copy(); o += 2 * ncoef
        dq = np.frombuffer(raw, np.int16, ncoef, o).copy(); o += 2 * ncoef
        rec = (np.frombuffer(raw, np.uint8, ny, o).reshape(H, W).copy(), np.frombuffer(raw, np.uint8, nc, o + ny).reshape(H // 2, W // 2).copy(),
               np.frombuffer(raw, np.uint8, nc, o + ny + nc).reshape(H // 2, W // 2).copy())
        o += ny + 2 * nc
        return dict(blocks=blocks, q=q, dq=dq, rec=rec)
    ref = side()
    brc = struct.unpack_from("<i", raw, o)[0]
    o += 4
    return ref, (side() if run_binding else None), brc
