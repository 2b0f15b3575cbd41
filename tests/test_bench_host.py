"""CPU: the host-side logic of bench.py -- the synthesised mode decision (the two mode-info grids from ME results), the mini-GOP
structure helpers for both hierarchies, and the multi-process cpu_baseline leg -- on small pictures: the grids must be well-formed
for the product's list builder, and the CPU leg must run whole pictures through every stage of the step."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

import encdec_model as M
import svt_testlib as T

B = T.B
_sp = importlib.util.spec_from_file_location("bench", os.path.join(T.ROOT, "bench.py"))
bench = importlib.util.module_from_spec(_sp)
sys.modules["bench"] = bench            # the multi-process leg pickles its worker by module name
_sp.loader.exec_module(bench)
if T.ROOT not in sys.path:
    sys.path.append(T.ROOT)

W, H = 256, 200   # 4 x 4 superblocks, the last SB row 8 samples high


def test_structure_helpers_both_hierarchies():
    for levels in (4, 3):
        bench.set_structure(levels)
        n = 1 << levels
        assert bench.MINIGOP == n and len(bench.LAYER) == n and bench.LAYER[-1] == 0 and max(bench.LAYER) == levels
        assert sorted(sum((bench.pics_of_layer(l) for l in range(levels + 1)), [])) == list(range(1, n + 1))
        for i in range(1, n + 1):
            a, b = bench.refs_of(i)
            if i == n:
                assert (a, b) == (0, 0)
            else:   # references are lower-layer pictures (or the previous base) on both sides
                assert a < i < b and all(j in (0,) or bench.LAYER[j - 1] < bench.LAYER[i - 1] for j in (a, b))
    bench.set_structure(4)


def test_synthesised_decision_is_wellformed_and_the_chain_runs():
    frames = T.gen_clip_subpel(W, H, 3, 7)
    pics = [T.PaPic(f) for f in frames]
    res, _ = T.oracle_me_picture(pics[1], pics[0], pics[2], B.me_params_preset(W, H, 9, 1, 2, 1, 4))
    kinds = bench.partition_kinds(np.random.default_rng(3), W, H)
    mi_rows, mi_cols, nsbx = H // 8, W // 8, (W + 63) // 64
    mi, k_cell = bench.build_mode_info(B, res, kinds, mi_rows, mi_cols, nsbx)
    lmi = bench.build_lf_mode_info(B, k_cell, mi_rows, mi_cols, 20)
    for r in range(mi_rows):      # every unit of a block carries the block's values; blocks are aligned to their size and inside the picture
        for c in range(mi_cols):
            bw = int(mi["bw8"][r, c])
            r0, c0 = r - r % bw, c - c % bw
            assert mi[r0, c0] == mi[r, c] and r0 + bw <= mi_rows and c0 + bw <= mi_cols
            assert lmi["sb_type"][r, c] == {1: 3, 2: 6, 4: 9}[bw] and lmi["tx_size"][r, c] == k_cell[r, c]
    assert set(np.unique(mi["ref_list"][..., 0]).tolist()) <= {0, 1} and (mi["ref_list"][..., 1] >= -1).all()
    z = res[0, 1]   # SB 0, first 32x32 PU: MVs are the PU's ME result in 1/8 sample
    if kinds[0, 0] == 3:
        assert int(mi["mv_col"][0, 0, 0]) == 2 * int(z["x_mv_l1"] if z["dir0"] == 1 else z["x_mv_l0"])
    # the oracle chain accepts the grids (the product's list builder rejects malformed ones) and reconstructs the picture
    chroma = lambda y, k: ((y[::2, ::2] // 2 + 32).astype(np.uint8), (255 - y[::2, ::2] // 2).astype(np.uint8))
    refs = [M.RefPic(W, H).set_padded(frames[k], *chroma(frames[k], k)) for k in (0, 2)]
    fl = B.EncdecFlags(limit_intra=0, allow_enc_dec_mismatch=0, do_recon=1, apply_loop_filter=1, pad_reference=1)
    thr = B.LfThresh()
    B.load().svt_hip_lf_thresh_init(C.byref(thr), 0)
    tm = {}
    o = M.oracle_encdec_picture((frames[1],) + chroma(frames[1], 1), refs, mi, lmi, bench.Q_INDEX, fl, thr, timings=tm)
    assert set(tm) == {"mc", "lists", "tq", "skip", "lf", "pad"}
    assert np.mean(np.abs(o["rec"].interior()[0].astype(int) - frames[1])) < 10
    n0, n1, n2, n3 = (int((k_cell == k).sum()) // d for k, d in ((0, 1), (1, 1), (2, 4), (3, 16)))
    assert o["counts"] == [6 * n0 + 2 * n1, n1 + 2 * n2, n2 + 2 * n3, n3]      # the block counts bench.py reports as workload statistics


def test_cpu_baseline_leg_runs_one_process_per_picture():
    """the whole cpu_baseline leg (native oracle build, worker processes, every stage) on a 17-picture clip of small pictures"""
    bench.set_structure(4)
    frames = T.gen_clip(W, H, bench.MINIGOP + 1, 11)
    src_all = np.zeros((bench.MINIGOP + 1, W * H * 3 // 2), np.uint8)
    for i, y in enumerate(frames):
        src_all[i, :W * H] = y.ravel()
        src_all[i, W * H:W * H * 5 // 4] = (y[::2, ::2] // 2 + 32).ravel()
        src_all[i, W * H * 5 // 4:] = (255 - y[::2, ::2] // 2).ravel()
    mi_rows, mi_cols, nsbx = H // 8, W // 8, (W + 63) // 64
    rng = np.random.default_rng(1)
    mc, lf = {}, {}
    res0 = np.zeros((T.n_sb(W, H), 85), B.ME_RESULT_DTYPE)
    for i in range(1, bench.MINIGOP + 1):
        kinds = bench.partition_kinds(rng, W, H)
        mc[i], k_cell = bench.build_mode_info(B, res0, kinds, mi_rows, mi_cols, nsbx)
        lf[i] = bench.build_lf_mode_info(B, k_cell, mi_rows, mi_cols, 20)
    out = bench.cpu_baseline(T, B, frames, src_all, mc, lf, None, W, H, 8, 1, True)
    assert out["kind"] == "port" and out["value"] > 0 and out["value_8_cores"] > 0 and 1 <= out["cores"] <= (os.cpu_count() or 1)
    assert out["cores"] == bench.effective_cpus()[0] or out["cores"] < bench.effective_cpus()[0]
    assert set(out["stage_seconds_per_picture_1_core"]) == {"pa", "me", "mc", "lists", "tq", "skip", "lf", "pad"} and out["cpu_model"]
