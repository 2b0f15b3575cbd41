"""tools/r06_api_c5.py [pictures] -- the public-API path at BASELINE C5's parameters (2160p, enc-mode 3, tune 0) under the stream knobs"""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svt_testlib as T
W, H, K = 3840, 2160, 17
N = int(sys.argv[1]) if len(sys.argv) > 1 else 260
path = os.path.join(tempfile.gettempdir(), "clip_c5.yuv")
with open(path, "wb") as f:
    for y in T.gen_clip(W, H, K, 5):
        y = np.ascontiguousarray(y); f.write(y.tobytes()); f.write((y[::2, ::2] // 2 + 32).astype(np.uint8).tobytes()); f.write(np.full((H // 2, W // 2), 128, np.uint8).tobytes())
for name, extra in (("default", {}), ("ME stream", {"SVT_HIP_ME_STREAM": "1"}), ("ME + deep stream", {"SVT_HIP_ME_STREAM": "1", "SVT_HIP_DEEP_STREAM": "1"})):
    for mode, tune in ((3, 0), (8, 1)):
        for recon in ("0", "1"):
            env = dict(os.environ); env.update(extra)
            vals = []
            for _ in range(2):
                r = subprocess.run([os.path.join(ROOT, "app", "svt_enc_api_bench"), path, str(W), str(H), str(K), str(N), str(mode), str(tune), recon], capture_output=True, text=True, env=env)
                vals.append(round(json.loads(r.stdout.strip().splitlines()[-1])["frames_per_s"]) if r.returncode == 0 else 0)
            print(f"{name:18s} enc-mode {mode} recon {recon}: {max(vals)} {vals}", flush=True)
