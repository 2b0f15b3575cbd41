"""Scope row f-3: the REFERENCE's own sample application (Source/App/*.c compiled as the files lie by oracle/Makefile, target
_ref/SvtVp9EncApp_on_shim) linked against this repository's libSvtVp9Enc.so -- the drop-in claim made with the reference's own
caller: it parses its command line, drives init / set_parameter / init_encoder / send_picture / get_packet / get_recon exactly as it
drives the reference library (App/EbAppContext.c:355-427, App/EbAppProcessCmd.c:437-760), writes its IVF container around the
(zero-byte: no entropy coder behind this path) packets and, with -o, the reconstructed pictures -- which must equal the oracle chain.
The binary is built in the build container and travels to the GPU box prebuilt; no GPU test reads /root/reference."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import svt_testlib as T
from test_enc_shim_encdec import chroma, oracle_clip, structure

B = T.B
APP = os.path.join(T.REF_DIR, "SvtVp9EncApp_on_shim")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(APP), reason="oracle/_ref/SvtVp9EncApp_on_shim not built (reference absent at build time)")]


def run_app(W, H, N, enc_mode, tune, qp, with_recon, td, seed=41):
    frames = T.gen_clip_subpel(W, H, N, seed)
    src = os.path.join(td, "in.yuv")
    with open(src, "wb") as f:
        for n, y in enumerate(frames):
            u, v = chroma(y, n)
            f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())
    ivf, rec = os.path.join(td, "out.ivf"), os.path.join(td, "recon.yuv")
    cmd = [APP, "-i", src, "-w", str(W), "-h", str(H), "-n", str(N), "-fps", "60", "-enc-mode", str(enc_mode), "-tune", str(tune), "-q", str(qp), "-b", ivf]
    if with_recon:
        cmd += ["-o", rec]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    return frames, r, ivf, rec


def check_ivf(path, W, H, N):
    raw = open(path, "rb").read()
    assert len(raw) == 32 + 12 * N                                   # stream header + N frame headers with empty payloads
    want = struct.pack("<4sHHIHHIIII", b"DKIF", 0, 32, 0x30395056, W, H, 60 * 1000, 1000, 0, 0)   # write_ivf_stream_header (:515-540)
    assert raw[:32] == want
    lib = B.load()
    import ctypes as C
    hdr = C.create_string_buffer(32)
    assert lib.svt_ivf_stream_header(hdr, W, H, 60 << 16, 0, 0) == 0 and hdr.raw == raw[:32]      # the product's own writer agrees
    pts = []
    for k in range(N):
        size, p = struct.unpack_from("<IQ", raw, 32 + 12 * k)
        assert size == 0
        pts.append(p)
    assert sorted(pts) == list(range(N))
    return pts


def test_reference_app_360p_recon_equals_oracle_chain():
    W, H, N, enc_mode, tune, qp = 640, 360, 20, 9, 1, 40
    with tempfile.TemporaryDirectory() as td:
        frames, r, ivf, rec = run_app(W, H, N, enc_mode, tune, qp, True, td)
        assert r.returncode == 0, (r.returncode, r.stdout[-400:], r.stderr[-400:])
        assert "Encoder finished" in r.stdout and f"{N:12d}" in r.stdout           # its own summary: Total Frames
        check_ivf(ivf, W, H, N)
        got = np.fromfile(rec, np.uint8)
    assert got.size == N * W * H * 3 // 2
    recs, outs = oracle_clip(frames, W, H, N, enc_mode, tune, qp, 1, 64, False)       # -intra-period default -2 -> 64 at 60 frames/s
    for k in range(N):
        y, u, v = recs[k].interior()
        want = np.concatenate([y.ravel(), u.ravel(), v.ravel()])
        assert np.array_equal(got[k * want.size:(k + 1) * want.size], want), k


def test_reference_app_4k():
    W, H, N, enc_mode, tune, qp = 3840, 2160, 18, 8, 1, 40
    with tempfile.TemporaryDirectory() as td:
        frames, r, ivf, rec = run_app(W, H, N, enc_mode, tune, qp, True, td, seed=11)
        assert r.returncode == 0, (r.returncode, r.stdout[-400:], r.stderr[-400:])
        assert "Encoder finished" in r.stdout
        check_ivf(ivf, W, H, N)
        pic = W * H * 3 // 2
        got = np.memmap(rec, np.uint8, mode="r")
        assert got.size == N * pic
        recs, outs = oracle_clip(frames, W, H, N, enc_mode, tune, qp, 1, 64, False)
        for k in range(N):
            y, u, v = recs[k].interior()
            want = np.concatenate([y.ravel(), u.ravel(), v.ravel()])
            assert np.array_equal(got[k * pic:(k + 1) * pic], want), k
