"""Multi-process (gloo, world_size 2, CPU) test of the GOP-shard path used by `bench.py --gpus N`.

Each rank runs the oracle ME on the GOPs assigned to it; the gathered per-GOP results must equal a serial run over all
GOPs, in presentation order, and the elapsed-time reduction must return the slowest rank's time.
"""
import importlib.util
import os
import socket
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
N_GOPS = 4


def _load_shard():
    spec = importlib.util.spec_from_file_location("gop_shard", os.path.join(ROOT, "svt-vp9_amd", "gop_shard.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _gop_result(gop):
    """CRC of the oracle's ME results for the B picture of a tiny 3-picture GOP."""
    sys.path.insert(0, HERE)
    import me_configs as MC
    import svt_testlib as T
    S = _load_shard()
    frames = T.gen_clip(136, 72, 3, S.gop_seed(40, gop))
    pics = [T.PaPic(f) for f in frames]
    res, _rcme = T.oracle_me_picture(pics[1], pics[0], pics[2], MC.preset("c1_360p_m9", 2, 1))
    return zlib.crc32(np.ascontiguousarray(res).view(np.uint8).tobytes())


def _worker(rank, world, port, out_path):
    import json
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = _load_shard()
    mine = S.assign_gops(N_GOPS, world)[rank]
    out = [(g, _gop_result(g)) for g in mine]
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    dist.barrier()
    slowest = S.reduce_elapsed(1.0 + rank, dist)          # rank r pretends to have taken 1+r seconds
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"merged": S.merge_in_presentation_order(gathered, world), "slowest": slowest,
                       "rate": S.aggregate_rate(16, 3, world, slowest)}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_assign_gops_round_robin():
    S = _load_shard()
    assert S.assign_gops(5, 2) == [[0, 2, 4], [1, 3]]
    assert S.assign_gops(3, 8)[:4] == [[0], [1], [2], []]
    assert S.merge_in_presentation_order([[(0, "a"), (2, "c")], [(1, "b")]], 2) == ["a", "b", "c"]
    assert S.reduce_elapsed(2.5) == 2.5 and S.aggregate_rate(16, 10, 8, 2.0) == 640.0


@pytest.mark.timeout(240)
def test_gop_shard_world2_gloo(tmp_path):
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "rank0.json")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), "2", str(port), out]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=200) == 0
    got = json.load(open(out))
    assert got["merged"] == [_gop_result(g) for g in range(N_GOPS)]   # same results, presentation order
    assert got["slowest"] == 2.0                                       # MAX over ranks
    assert got["rate"] == 16 * 3 * 2 / 2.0


if __name__ == "__main__" and len(sys.argv) == 6 and sys.argv[1] == "--worker":
    _worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
