"""Developer tool: per-workgroup start/end timeline of one frame (needs a -DRM_TIMELINE build
selected with RMHIP_LIB)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
w, h = int(sys.argv[1]) if len(sys.argv) > 1 else 1920, int(sys.argv[2]) if len(sys.argv) > 2 else 1080
mat = Material(); rm = Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms())
mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
mat.set_kernel(int(os.environ.get("RM_KERNEL", "0")))
mat.SetFloat("_RmStaticFraction", float(os.environ.get("RM_FRAC", "0.7")))
rt = RenderTexture(w, h, device="cuda:0")
for _ in range(20): rm.OnRenderImage(None, rt)
torch.cuda.synchronize()
wps, ns = C.c_int(), C.c_int()
mat._ck(mat._lib.rm_debug_counter_slots(mat.handle, None, 0, C.byref(wps), C.byref(ns)))
buf = np.zeros(wps.value * ns.value, dtype=np.uint32)
mat._ck(mat._lib.rm_debug_counter_slots(mat.handle, buf.ctypes.data_as(C.POINTER(C.c_uint32)), buf.size, None, None))
s = buf.reshape(ns.value, wps.value)
assert wps.value == 16, "not a timeline build"
t0 = s[:, 3].astype(np.uint64) | (s[:, 4].astype(np.uint64) << np.uint64(32))
t1 = s[:, 5].astype(np.uint64) | (s[:, 6].astype(np.uint64) << np.uint64(32))
base = t0.min()
st, en = (t0 - base).astype(np.float64) / 100.0, (t1 - base).astype(np.float64) / 100.0      # us
print(f"blocks {ns.value}; frame span {en.max():.1f} us; last block start {st.max():.1f} us")
dur = en - st
print("block duration us: mean %.2f  p50 %.2f  p99 %.2f  max %.2f" % (dur.mean(), np.median(dur), np.percentile(dur, 99), dur.max()))
i = int(np.argmax(en)); print("last block to end: id %d start %.1f end %.1f evals p/s %d/%d" % (i, st[i], en[i], s[i, 1], s[i, 2]))
j = np.argsort(dur)[::-1][:8]
for b in j: print("  long block id %6d (row %3d col %3d) start %6.1f end %6.1f dur %6.1f" % (b, b // ((w + 15) // 16), b % ((w + 15) // 16), st[b], en[b], dur[b]))
print("per-slot tiles done: mean %.1f max %d; max tile cost seen %d" % ((s[:,7]&0xffff).mean(), (s[:,7]&0xffff).max(), (s[:,7]>>16).max()))
print("wave end-time percentiles 10/50/90/99/100:", np.round(np.percentile(en,[10,50,90,99,100]),1))
hv=(s[:,7]>>16)>=40
print("waves that saw a heavy tile:", int(hv.sum()), "their end times p50/max: %.1f %.1f" % (np.median(en[hv]), en[hv].max()), "; others p50/max: %.1f %.1f" % (np.median(en[~hv]), en[~hv].max()))
if os.environ.get("RM_KERNEL")=="2": print("claims per wave mean %.1f; wait per claim mean %.2f us; wait per wave mean %.1f us max %.1f us" % (s[:,1].mean(), s[:,0].sum()/max(1,s[:,1].sum())/100.0, s[:,0].mean()/100.0, s[:,0].max()/100.0))
edges = np.arange(0, en.max() + 5, 5.0)
resident = [(int(((st <= t) & (en > t)).sum())) for t in edges]
print("resident blocks every 5 us:", resident)
started = [int((st <= t).sum()) for t in edges]
print("started blocks every 5 us:", started)
late = np.argsort(en)[::-1][:16]
print("the waves that end last: end us, tiles done, longest tile; then their last four tiles as start us (in the frame) / cost in evaluations / queue entry")
for b in late:
    tiles = ["%6.1f/%3d/%6d" % (st[b] + s[b, 8 + 2 * k] / 100.0, s[b, 9 + 2 * k] >> 20, s[b, 9 + 2 * k] & 0xfffff) for k in range(4)]
    print("   %6.1f  %3d  %4d   %s" % (en[b], s[b, 7] & 0xffff, s[b, 7] >> 16, "  ".join(tiles)))
# how much work started in the last stretch of the frame
last_start = st + s[:, 14] / 100.0
print("waves whose last tile started in the last 30 / 20 / 10 us of the frame:", [int((last_start > en.max() - d).sum()) for d in (30, 20, 10)])
cost_last = s[:, 15] >> 20
late_mask = last_start > en.max() - 30
print("cost of those last tiles (evaluations): mean %.1f max %d; of all last tiles: mean %.1f" % (cost_last[late_mask].mean(), cost_last[late_mask].max(), cost_last.mean()))
