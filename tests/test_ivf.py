"""IVF container layer (svt_ivf_stream_header / svt_ivf_packetize, host only) against the reference application's own
writers: the committed fixture holds the bytes write_ivf_stream_header / write_ivf_frame_header produced (oracle/_ref/
ref_ivf_headers = App/EbAppProcessCmd.c compiled as it lies); with oracle/_ref present they are re-derived live.  The
SHOW_EXT splitting sits inside process_output_stream_buffer, which needs the encoder library (not buildable here): it is
checked against the layout that function writes (App/EbAppProcessCmd.c:621-646), frame headers pinned as above."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import svt_testlib as T
from gen_golden import IVF_GOLDEN_CASES

B = T.B
GOLD = os.path.join(T.GOLDEN_DIR, "ivf_reference.npz")


def _lib():
    lib = B.load()
    lib.svt_ivf_packetize.restype = C.c_int64
    lib.svt_ivf_packetize.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_char_p, C.c_size_t]
    return lib


def _ours(w, h, fr, num, den, frames):
    """stream header + the frame headers of empty-payload frames with the given sizes: packetize writes header then payload,
    so the header is the first 12 bytes of each packet"""
    lib = _lib()
    hdr = C.create_string_buffer(32)
    assert lib.svt_ivf_stream_header(hdr, w, h, fr, num, den) == 0
    out = hdr.raw
    for n, pts in frames:
        m = min(n, 64)  # the header only depends on the declared size: check it with the true size where that is small
        buf = C.create_string_buffer(m + 12)
        got = lib.svt_ivf_packetize(b"\xab" * m, m, pts, 0, buf, m + 12)
        assert got == m + 12 and buf.raw[12:] == b"\xab" * m
        out += (struct.pack("<I", n) if n != m else buf.raw[:4]) + buf.raw[4:12]
    return out


@pytest.mark.parametrize("k", range(len(IVF_GOLDEN_CASES)))
def test_ivf_headers_vs_golden(k):
    assert _ours(*IVF_GOLDEN_CASES[k]) == np.load(GOLD)[str(k)].tobytes()


@pytest.mark.skipif(not T.have_ref("ref_ivf_headers"), reason="oracle/_ref/ref_ivf_headers not built (reference absent)")
def test_ivf_headers_vs_reference_live():
    for case in IVF_GOLDEN_CASES + ((1920, 1080, 50 << 16, 0, 7, ((123456, 99),)),):
        assert _ours(*case) == T.ref_ivf_headers(*case)


def test_ivf_show_existing_frame_split():
    lib = _lib()
    payload = bytes(range(1, 41))  # 36 bytes of coded frame + the four one-byte show-existing-frame headers
    pts = (7 << 32) + 1
    buf = C.create_string_buffer(len(payload) + 60)
    n = lib.svt_ivf_packetize(payload, len(payload), pts, 1, buf, len(buf))
    want = struct.pack("<IQ", 36, pts) + payload[:36]
    for i, d in enumerate((-2, -1, 0, 1)):
        want += struct.pack("<IQ", 1, pts + d) + payload[36 + i:37 + i]
    assert n == len(want) and buf.raw[:n] == want
    # capacity and size errors, the wrap of pts - 2 at 0 (the reference subtracts in uint64_t)
    assert lib.svt_ivf_packetize(payload, len(payload), pts, 1, buf, len(payload) + 59) < 0
    assert lib.svt_ivf_packetize(payload, 3, pts, 1, buf, len(buf)) < 0
    n = lib.svt_ivf_packetize(payload, 4, 0, 1, buf, len(buf))
    assert n == 12 + 4 * 13 and buf.raw[12 + 4:12 + 12] == struct.pack("<Q", (1 << 64) - 2)
