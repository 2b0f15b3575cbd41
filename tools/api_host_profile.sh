#!/bin/bash
# tools/api_host_profile.sh -- the public-API path (app/svt_enc_api_bench, 2160p, 130 and 600 pictures, best of 4) under a list of environment
# settings, with the host-time profile of the upload (SVT_HIP_SHIM_PROFILE=1).  Edit the `for extra in [...]` list for a sweep; RECON=1: with
# every reconstruction fetched.  Outputs of the round: profiles/r05_api_sweeps.txt.
# API path host profile: one run of 600 pictures, feeder on, 4 copy threads
python - <<'PY'
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svt_testlib as T
W, H, n_frames = 3840, 2160, 17
frames = T.gen_clip(W, H, n_frames, 5)
exe = os.path.join(ROOT, "app", "svt_enc_api_bench")
path = "/tmp/clip.yuv"
with open(path, "wb") as f:
    for y in frames:
        y = np.ascontiguousarray(y)
        f.write(y.tobytes()); f.write((y[::2, ::2] // 2 + 32).astype(np.uint8).tobytes()); f.write(np.full((H // 2, W // 2), 128, np.uint8).tobytes())
for extra in [{}, {"SVT_HIP_INPUT_STREAMS": "2"}, {}, {"SVT_HIP_INPUT_STREAMS": "2"}, {"SVT_HIP_INPUT_STREAMS": "2", "SVT_HIP_COPY_THREADS": "8"}]:
    env = dict(os.environ); env["SVT_HIP_SHIM_PROFILE"] = "1"; env.update(extra)
    for n in (130, 600):
        best = None
        vals = []
        for rep in range(4):
            r = subprocess.run([exe, path, str(W), str(H), str(n_frames), str(n), "8", "1", os.environ.get("RECON", "0")], capture_output=True, text=True, env=env)
            try: v = json.loads(r.stdout.strip().splitlines()[-1])["frames_per_s"]
            except Exception: v = 0.0; print("FAILED", r.stderr[-300:])
            vals.append(round(v))
            if best is None or v > best[0]: best = (v, [l for l in r.stderr.splitlines() if "upload profile" in l])
        print(extra, n, best[0], vals, (best[1] or [""])[-1].split("):")[-1], flush=True)
PY
