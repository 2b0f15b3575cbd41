"""Self-consistency stress of the public API's stream plumbing at 2160p: the reference's own application on libSvtVp9Enc.so codes the same
clip with everything on one stream (SVT_HIP_SINGLE_STREAM=1: plain stream order) and with the input / output contexts beside the main
one; the reconstructed files must be byte-identical.  `python tools/api_stream_ab.py [n_pictures] [intra_period]` on the GPU box."""
import filecmp
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svt_testlib as T

W, H = 3840, 2160
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ip = sys.argv[2] if len(sys.argv) > 2 else "39"
APP = os.path.join(ROOT, "oracle", "_ref", "SvtVp9EncApp_on_shim")
frames = T.gen_clip(W, H, 12, 5)
with tempfile.TemporaryDirectory() as td:
    src = os.path.join(td, "clip.yuv")
    with open(src, "wb") as f:
        for i in range(N):
            y = np.ascontiguousarray(frames[i % 12 if (i // 12) % 2 == 0 else 11 - i % 12])
            f.write(y.tobytes())
            f.write((y[::2, ::2] // 2 + 32).astype(np.uint8).tobytes())
            f.write((255 - y[1::2, ::2] // 2).astype(np.uint8).tobytes())
    outs = []
    for single in ("1", "0", "0"):
        rec = os.path.join(td, f"rec{len(outs)}.yuv")
        cmd = [APP, "-i", src, "-w", str(W), "-h", str(H), "-n", str(N), "-fps", "60", "-enc-mode", "8", "-tune", "1", "-q", "40", "-intra-period", ip,
               "-b", os.path.join(td, "o.ivf"), "-o", rec]
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, SVT_HIP_SINGLE_STREAM=single))
        assert r.returncode == 0, (r.stdout + r.stderr)[-400:]
        assert os.path.getsize(rec) == N * W * H * 3 // 2, os.path.getsize(rec)
        outs.append(rec)
    same = [filecmp.cmp(outs[0], o, shallow=False) for o in outs[1:]]
    print("pictures", N, "intra period", ip, "multi-stream reconstruction == single-stream reconstruction:", same, flush=True)
    sys.exit(0 if all(same) else 1)
