"""Instruction counts per ME phase: launches the ME kernel (4K, enc-mode 8, temporal layer 4, 2 pictures) 15 times, stopping
after phase k of the first list (library built with -DME_FINE_PROF).  Run under
`rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES` and difference the dispatches."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init()
import me_configs as MC, svt_testlib as T
B = T.B; lib = B.load()
ctx = C.c_void_p(); B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
W, H = (1920, 1080) if os.environ.get("ME_PRESET") == "c2" else (3840, 2160)
frames = T.gen_clip(W, H, 3, 11)
dev = torch.device("cuda", 0); keep = []
def desc(luma):
    pa = T.PaPic(luma); d = B.PaPicture()
    for name, (a, pad) in zip(("full", "quarter", "sixteenth"), pa.planes()):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev); keep.append(t)
        pl = B.Plane(); pl.buf = t.data_ptr(); pl.stride = t.shape[1]; pl.origin_x = pl.origin_y = pad
        pl.width, pl.height = t.shape[1] - 2 * pad, t.shape[0] - 2 * pad; setattr(d, name, pl)
    return d
d = [desc(f) for f in frames]
res = torch.zeros((T.n_sb(W, H), 850), dtype=torch.int32, device=dev)
p = (MC.preset_c5(2, int(os.environ.get("ME_TL", "3"))) if os.environ.get("ME_PRESET") == "c5" else
     MC.preset("c2_1080p_m8", 2, int(os.environ.get("ME_TL", "4")), 4) if os.environ.get("ME_PRESET") == "c2" else MC.preset("c3_2160p_m8", 2, int(os.environ.get("ME_TL", "4")), 4))
for kv in filter(None, os.environ.get("ME_HACK", "").split(",")):   # timing experiments: override parameter fields (results change)
    k_, v_ = kv.split("="); setattr(p, k_, int(v_))
STOPS = [int(x) for x in os.environ.get("ME_STOPS", "0,1,19,20,21,2,3,4,5,6,7,8,9,10,11,12,-1").split(",")]
NP = int(os.environ.get("ME_PICS", "1")); REP = int(os.environ.get("ME_REPS", "1"))
cur, r0, r1 = (B.PaPicture * NP)(*([d[1]] * NP)), (B.PaPicture * NP)(*([d[0]] * NP)), (B.PaPicture * NP)(*([d[2]] * NP))
rp = (C.c_void_p * NP)(*([res.data_ptr()] * NP))
for k in STOPS:
    os.environ["SVT_HIP_ME_STOP"] = str(k)
    for _ in range(REP):
        B.check(lib.svt_hip_me_batch_device(ctx, NP, cur, r0, r1, C.byref(p), rp, None))
        B.check(lib.svt_hip_ctx_synchronize(ctx))
print("ok")
