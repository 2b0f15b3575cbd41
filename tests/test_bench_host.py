"""CPU: the host-side logic of bench.py -- the synthesised mode decision (mode-info grids from ME results, transform block
lists, loop-filter mode info) and the threaded cpu_baseline leg -- on a small picture, stage by stage through the oracle:
the data one builder writes must be what the next stage reads."""
import ctypes as C
import importlib.util
import os

import numpy as np

import svt_testlib as T

B = T.B
_sp = importlib.util.spec_from_file_location("bench", os.path.join(T.ROOT, "bench.py"))
bench = importlib.util.module_from_spec(_sp)
_sp.loader.exec_module(bench)

W, H = 256, 200   # 4 x 4 superblocks, the last SB row 8 samples high


def _setup():
    frames = T.gen_clip_subpel(W, H, 3, 7)
    pics = [T.PaPic(f) for f in frames]
    p = B.me_params_preset(W, H, 9, 1, 2, 1, 4)
    res, _ = T.oracle_me_picture(pics[1], pics[0], pics[2], p)
    rng = np.random.default_rng(3)
    kinds = rng.integers(0, 4, ((H + 31) // 32, (W + 31) // 32))
    kinds[-1] = np.minimum(kinds[-1], 1)
    return frames, res, kinds


def test_partition_covers_every_sample_once_and_chain_runs():
    frames, res, kinds = _setup()
    mi_rows, mi_cols, nsbx = H // 8, W // 8, (W + 63) // 64
    mi, k_cell = bench.build_mode_info(B, res, kinds, mi_rows, mi_cols, nsbx)
    # every unit of a block carries the block's values; blocks are aligned to their size
    for r in range(mi_rows):
        for c in range(mi_cols):
            bw = int(mi["bw8"][r, c])
            r0, c0 = r - r % bw, c - c % bw
            assert mi[r0, c0] == mi[r, c] and r0 + bw <= mi_rows
    assert set(np.unique(mi["ref_list"][..., 0]).tolist()) <= {0, 1} and (mi["ref_list"][..., 1] >= -1).all()
    # MVs are the PU's ME result in 1/8 sample
    z = res[0, 1]   # SB 0, first 32x32 PU
    if kinds[0, 0] == 3:
        assert int(mi["mv_col"][0, 0, 0]) == 2 * int(z["x_mv_l1"] if z["dir0"] == 1 else z["x_mv_l0"])
    iscan, ioffs = T.iscan_array()
    arrs = bench.build_tq_blocks(B, kinds, W, H, W, [ioffs[(ts, 0)] for ts in range(4)])
    cover = np.zeros((H + H // 2, W), np.int32)
    for ts, a in enumerate(arrs):
        n = 4 << ts
        for off in a["src_off"]:
            y, x = divmod(int(off), W)
            cover[y:y + n, x:x + n] += 1
    assert (cover == 1).all()
    # chain: oracle MC on the grid -> oracle TQ on the block list -> LF mode info -> product mask builder -> oracle LF
    src = np.zeros((H + H // 2, W), np.uint8)
    y = frames[1]
    src[:H], src[H:, :W // 2], src[H:, W // 2:] = y, y[::2, ::2] // 2 + 32, 255 - y[::2, ::2] // 2
    pad = 80
    refs = []
    for k in (0, 2):
        yy = frames[k]
        refs.append(tuple(np.ascontiguousarray(np.pad(pl, pd, mode="edge")) for pl, pd in ((yy, pad), (yy[::2, ::2] // 2 + 32, pad // 2), (255 - yy[::2, ::2] // 2, pad // 2))))
    pr = T.oracle_mc_frame(dict(mi=mi, mi_rows=mi_rows, mi_cols=mi_cols, refs=refs, pad=pad, use_subpel=1, width=W, height=H))
    pred = np.zeros_like(src)
    pred[:H], pred[H:, :W // 2], pred[H:, W // 2:] = pr
    assert np.mean(np.abs(pred[:H].astype(int) - src[:H])) < 12    # the grid's MVs predict the picture
    blocks = np.concatenate(arrs)
    nn = 16 << (2 * blocks["tx_size"].astype(np.int64))
    blocks["coeff_off"] = np.concatenate([[0], np.cumsum(nn)[:-1]])
    lib = B.load()
    qrow = np.load(os.path.join(T.GOLDEN_DIR, "quant_reference.npz"))["0|0|0"][bench.Q_INDEX]
    qtabs = np.zeros(2, dtype=B.QUANT_DTYPE)
    for j, base in enumerate((2, 14)):
        assert lib.svt_hip_quant_tables_init(bench.Q_INDEX, int(qrow[1]), int(qrow[base]), int(qrow[base + 1]), qtabs[j:j + 1].ctypes.data_as(C.c_void_p)) == 0
    case = dict(src=src, pred=pred, blocks=blocks, counts=np.array([len(a) for a in arrs], np.int32), qtabs=qtabs, iscan=iscan, n_coeff=int(nn.sum()))
    recon, q, dq, eob = T.oracle_tq_batch(case)
    assert np.mean(np.abs(recon.astype(int) - src)) <= np.mean(np.abs(pred.astype(int) - src))
    luma = blocks["qtab"] == 0
    nz4 = np.zeros(((H + 31) // 32 * 8, (W + 31) // 32 * 8), bool)
    for b, e in zip(blocks[luma], eob[luma]):
        if e:
            yy, xx = divmod(int(b["recon_off"]), W)
            n4 = 1 << int(b["tx_size"])
            nz4[yy // 4:yy // 4 + n4, xx // 4:xx // 4 + n4] = True
    lmi = bench.build_lf_mode_info(B, k_cell, nz4, mi_rows, mi_cols, 20)
    assert lmi["skip"].min() == 0 and set(np.unique(lmi["sb_type"]).tolist()) <= {3, 6, 9}
    lfm_o, lfm_p = T.oracle_lf_build_masks(lmi, mi_rows, mi_cols), T.product_lf_build_masks(lmi, mi_rows, mi_cols)
    assert all(np.array_equal(lfm_o[n], lfm_p[n]) for n in lfm_o.dtype.names)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    lcase = dict(y=recon[:H].copy(), u=recon[H:, :W // 2].copy(), v=recon[H:, W // 2:].copy(), lfm=lfm_p, thr=thr, mi_rows=mi_rows, mi_cols=mi_cols)
    out = T.oracle_lf_frame(lcase)
    assert (out[0] != lcase["y"]).any()


def test_cpu_baseline_leg_runs_threaded():
    """the whole cpu_baseline leg (native oracle build, six stages, thread pools) on a 17-picture clip of small pictures"""
    frames = T.gen_clip(W, H, bench.MINIGOP + 1, 11)
    src_all = np.zeros((bench.MINIGOP + 1, H + H // 2, W), np.uint8)
    for i, y in enumerate(frames):
        src_all[i, :H], src_all[i, H:, :W // 2], src_all[i, H:, W // 2:] = y, y[::2, ::2] // 2 + 32, 255 - y[::2, ::2] // 2
    mi_rows, mi_cols, nsbx = H // 8, W // 8, (W + 63) // 64
    iscan, ioffs = T.iscan_array()
    rng = np.random.default_rng(1)
    mi_list, by_ts = [], [[] for _ in range(4)]
    pics = [T.PaPic(f) for f in frames]
    for i in range(1, bench.MINIGOP + 1):
        a, b = bench.refs_of(i)
        res, _ = T.oracle_me_picture(pics[i], pics[a], pics[b], B.me_params_preset(W, H, 8, 1, 2, bench.LAYER[i - 1], 4)) if i in (8, 4, 12, 2) else (np.zeros((T.n_sb(W, H), 85), B.ME_RESULT_DTYPE), None)
        kinds = rng.integers(0, 4, ((H + 31) // 32, (W + 31) // 32))
        kinds[-1] = np.minimum(kinds[-1], 1)
        mi_list.append(bench.build_mode_info(B, res, kinds, mi_rows, mi_cols, nsbx)[0])
        for ts, arr in enumerate(bench.build_tq_blocks(B, kinds, W, H, W, [ioffs[(ts, 0)] for ts in range(4)])):
            for f in ("src_off", "pred_off", "recon_off"):
                arr[f] += np.uint32((i - 1) * (H + H // 2) * W)
            arr["src_off"] += np.uint32((H + H // 2) * W)
            by_ts[ts].append(arr)
    blocks = np.concatenate([a for ts in range(4) for a in by_ts[ts]])
    pic_of = np.concatenate([np.full(len(a), k, np.int32) for ts in range(4) for k, a in enumerate(by_ts[ts])])
    nn = 16 << (2 * blocks["tx_size"].astype(np.int64))
    blocks["coeff_off"] = np.concatenate([[0], np.cumsum(nn)[:-1]])
    qtabs = np.array([T.quant_table(223, 305)] * 2, dtype=B.QUANT_DTYPE)
    rtab, rscan = T.rate_tables()
    roffs, _ = T.rate_scan_offsets()
    rb = np.zeros(len(blocks), dtype=B.RATE_BLOCK_DTYPE)
    rb["coeff_off"], rb["tx_size"], rb["is_inter"], rb["plane_type"] = blocks["coeff_off"], blocks["tx_size"], 1, blocks["qtab"]
    rb["scan_off"] = np.array([roffs[(ts, 0)] for ts in range(4)], np.uint32)[blocks["tx_size"]]
    sb_rows, sb_cols = (mi_rows + 7) // 8, (mi_cols + 7) // 8
    lfms = [T.gen_lf_masks(np.random.default_rng(k), sb_rows, sb_cols) for k in range(bench.MINIGOP)]
    thr = B.LfThresh()
    B.load().svt_hip_lf_thresh_init(C.byref(thr), 0)
    out = bench.cpu_baseline(T, B, frames, src_all, mi_list, blocks, pic_of, qtabs, iscan, rb, rtab, rscan, lfms, thr, W, H, W, True, int(nn.sum()))
    assert out["kind"] == "port" and out["cores"] == (os.cpu_count() or 1) and out["value"] > 0 and out["value_8_cores"] > 0
    assert set(out["stage_seconds_all_cores"]) == set(bench.STAGES) and out["cpu_model"]
