"""Randomised parity runs of the picture-level EncDec driver (svt_hip_encdec_batch_device) on the GPU against the oracle chain: picture
sizes, batch sizes, q indices, the reference's flag combinations, shares of intra blocks (incl. 4x4 units) inside the inter pictures.
`python tools/encdec_fuzz.py [n_cases] [seed]` on the GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np

import encdec_model as M
import svt_testlib as T
import test_gpu_encdec as TE

B = T.B
lib = B.load()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = C.c_void_p()
B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
thr = B.LfThresh()
lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
for k in range(n_cases):
    W, H = 8 * int(rng.integers(9, 40)), 8 * int(rng.integers(9, 28))
    n_pics = int(rng.integers(1, 6))
    q_index = int(rng.integers(1, 256))
    cfg = dict(enc_mode=int(rng.integers(0, 10)), tune=int(rng.integers(0, 3)), temporal_layer_index=int(rng.integers(0, 5)), is_used_as_reference=int(rng.integers(0, 2)),
               recon_file=int(rng.integers(0, 2)), loop_filter=int(rng.integers(0, 2)))
    flags = TE.flags_of(**cfg)
    share = float(rng.choice([0.0, 0.15, 0.5])) if flags.do_recon else 0.0
    srcs, refs, me = TE.make_inputs(W, H, n_pics, seed=int(rng.integers(1, 10000)))
    level = lib.svt_hip_lf_level_from_q(lib.svt_hip_vp9_ac_step(q_index), 0)
    grids, n_intra = [], 0
    for i, m in enumerate(me):
        mc, lf = TE.md_host(m, W, H, int(rng.integers(50, 2000)), level)
        if share:
            lf, mc, n = M.make_mixed(int(rng.integers(1, 10000)), lf, mc, share=share, level=level)
            n_intra += n
        grids.append((mc, lf))
    rec_inits = [M.RefPic(W, H) for _ in range(n_pics)]
    for r in rec_inits:
        r.buf[:] = rng.integers(0, 256, r.buf.size, dtype=np.uint8)
    dp, blocks, pos, eob, cnt = TE.run_device(ctx, W, H, srcs, refs, grids, q_index, flags, thr, rec_inits, has_intra=int(n_intra > 0))
    for i in range(n_pics):
        rec0 = M.RefPic(W, H)
        rec0.buf[:] = rec_inits[i].buf
        o = M.oracle_encdec_picture(srcs[i], refs, grids[i][0], grids[i][1], q_index, flags, thr, recon_init=rec0)
        d = dp[i]
        bad = []
        if not np.array_equal(d.q_t.cpu().numpy(), o["qcoeff"]) or not np.array_equal(d.dq_t.cpu().numpy(), o["dqcoeff"]):
            bad.append("coefficients")
        if not np.array_equal(d.emap_t.cpu().numpy().view(np.uint16), o["eob_map"]):
            bad.append("eob map")
        lf_g = d.lf_t.cpu().numpy().view(B.LF_MODE_INFO_DTYPE).reshape(H // 8, W // 8)
        if not np.array_equal(lf_g["skip"], o["lf_mi"]["skip"]):
            bad.append("skip")
        if flags.apply_loop_filter and not TE.masks_equal(d.lfm_t.cpu().numpy().view(B.LF_MASK_DTYPE).reshape(o["lfm"].shape), o["lfm"]):
            bad.append("masks")
        if not np.array_equal(d.rec_t.cpu().numpy(), o["rec"].buf):
            bad.append("reconstruction")
        if bad:
            print("FAIL", k, W, H, n_pics, q_index, cfg, share, "picture", i, bad, flush=True)
            sys.exit(1)
print(f"encdec fuzz: {n_cases} cases OK", flush=True)
