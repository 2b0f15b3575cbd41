"""Static instruction counts of the ME kernel per phase: compiles csrc/me_kernel.hip with -DME_ASM_MARKS (ME_MARK(i) leaves a
comment in the assembly) and counts the VALU / SALU / LDS / VMEM instructions between consecutive marks of instance SPEC
(default 1).  Static: a loop body counts once, blocks the compiler moved are attributed to where they landed -- an aid for the
instruction diet without a GPU; tools/me_phase_profile.sh gives the dynamic counts on the box."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
extra = sys.argv[2:]
out = "/tmp/me_static.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
                       "-DME_ASM_MARKS", "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "svt-vp9_amd/csrc/me_kernel.hip")] + extra,
                      stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
kname = os.environ.get("KERNEL", "svt_me_sb_kernel")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z%d%sILi%dE" % (len(kname), kname, spec)))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
cur = "entry"; cnt = collections.OrderedDict()
def cls(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    return None
detail = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    m = re.match(r"\s*; @MARK (\d+)", l)
    if m: cur = "after_mark_%s" % m.group(1); continue
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m: continue
    k = cls(m.group(1))
    if not k: continue
    cnt.setdefault(cur, collections.Counter())[k] += 1
    detail[cur][m.group(1)] += 1
tot = collections.Counter()
for k, v in cnt.items():
    print("%-16s valu %5d salu %5d lds %4d vmem %4d" % (k, v["valu"], v["salu"], v["lds"], v["vmem"])); tot.update(v)
print("%-16s valu %5d salu %5d lds %4d vmem %4d" % ("total", tot["valu"], tot["salu"], tot["lds"], tot["vmem"]))
if os.environ.get("DETAIL"):
    for k, v in detail.items():
        print(k, ", ".join("%s %d" % kv for kv in v.most_common(14)))
