"""host/copy_pool.c: the staging copy of an input picture (the copy eb_vp9_svt_enc_send_picture makes of the caller's planes,
Source/Lib/Codec/EbEncHandle.c:2743-2796), one fork-join over a few threads for all planes of the picture.  CPU only."""
import ctypes as C
import numpy as np
import pytest
import svt_testlib as T

B = T.B


def _lib():
    lib = B.load()
    lib.svt_copy_planes_mt.restype = None
    lib.svt_copy_rows_mt.restype = None
    return lib


@pytest.mark.parametrize("W,H", [(64, 64), (1920, 1080), (3840, 2160)])
def test_planes_copy_equals_numpy(W, H):
    lib = _lib()
    rng = np.random.default_rng(W + H)
    strides = (W + 48, W // 2 + 16, W // 2)          # padded luma / Cb, tight Cr
    shapes = ((H, W), (H // 2, W // 2), (H // 2, W // 2))
    src = [rng.integers(0, 256, (h, st), dtype=np.uint8) for (h, _), st in zip(shapes, strides)]
    dst_strides = (W, W // 2 + 8, W // 2)
    dst = [np.full((h, st), 7, dtype=np.uint8) for (h, _), st in zip(shapes, dst_strides)]
    P = C.c_void_p * 3
    S = C.c_size_t * 3
    lib.svt_copy_planes_mt(3, P(*[d.ctypes.data for d in dst]), S(*dst_strides), P(*[s.ctypes.data for s in src]), S(*strides), S(*[w for _, w in shapes]),
                           S(*[h for h, _ in shapes]))
    for d, s_, (h, w) in zip(dst, src, shapes):
        assert np.array_equal(d[:, :w], s_[:, :w])
        assert (d[:, w:] == 7).all()                 # nothing written beyond a row's width


def test_rows_copy_and_repeated_use():
    lib = _lib()
    rng = np.random.default_rng(3)
    for rep in range(6):                             # the pool is reused: generations must not get lost
        h, w, st = 1200 + rep, 2048, 2048 + 64 * (rep & 1)
        src = rng.integers(0, 256, (h, st), dtype=np.uint8)
        dst = np.zeros((h, w), dtype=np.uint8)
        lib.svt_copy_rows_mt(C.c_void_p(dst.ctypes.data), C.c_size_t(w), C.c_void_p(src.ctypes.data), C.c_size_t(st), C.c_size_t(w), C.c_size_t(h))
        assert np.array_equal(dst, src[:, :w])
