import sys, os, subprocess, numpy as np
sys.path.insert(0, "tests")
import svt_testlib as T
W, H = 3840, 2160
with open("/tmp/clip.yuv", "wb") as f:
    for y in T.gen_clip(W, H, 17, 5):
        y = np.ascontiguousarray(y); f.write(y.tobytes()); f.write((y[::2, ::2] // 2 + 32).astype(np.uint8).tobytes()); f.write(np.full((H // 2, W // 2), 128, np.uint8).tobytes())
env = dict(os.environ); env["SVT_HIP_SHIM_PROFILE"] = "2"
for n, rec in ((130, "0"), (130, "0"), (130, "0"), (600, "0"), (130, "1"), (600, "1")):
    r = subprocess.run(["app/svt_enc_api_bench", "/tmp/clip.yuv", "3840", "2160", "17", str(n), "8", "1", rec], capture_output=True, text=True, env=env)
    print(r.stdout.strip().splitlines()[-1]); print("   ", "; ".join(l.split(":")[0][10:] + l.split(":")[1][:10] for l in r.stderr.splitlines() if "key frame" in l)[:200]); print("   ", [l[38:] for l in r.stderr.splitlines() if "host time" in l][-1:])
