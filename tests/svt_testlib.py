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
