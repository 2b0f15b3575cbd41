"""A/B of the public-API path (app/svt_enc_api_bench): input side on its own stream (default) vs everything on one stream
(SVT_HIP_SINGLE_STREAM=1), 2160p, with and without reconstruction output.  Run on the GPU box."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svt_testlib as T

W, H, n_frames = 3840, 2160, 8
n_send = int(sys.argv[1]) if len(sys.argv) > 1 else 130
frames = T.gen_clip(W, H, n_frames, 5)
exe = os.path.join(ROOT, "app", "svt_enc_api_bench")
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "clip.yuv")
    with open(path, "wb") as f:
        for y in frames:
            y = np.ascontiguousarray(y)
            f.write(y.tobytes())
            f.write((y[::2, ::2] // 2 + 32).astype(np.uint8).tobytes())
            f.write(np.full((H // 2, W // 2), 128, np.uint8).tobytes())
    for recon in (0, 1):
        for single in ("1", "0", "1", "0"):
            env = dict(os.environ, SVT_HIP_SINGLE_STREAM=single)
            if len(sys.argv) > 2:
                env["SVT_HIP_COPY_THREADS"] = sys.argv[2]
            r = subprocess.run([exe, path, str(W), str(H), str(n_frames), str(n_send), "8", "1", str(recon)], capture_output=True, text=True, env=env)
            d = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": (r.stdout + r.stderr)[-300:]}
            print("recon", recon, "single_stream", single, d.get("frames_per_s"), d.get("seconds"), d.get("error", ""), flush=True)
