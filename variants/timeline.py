"""Wave timeline of one kernel of the fused call on the bench scene (ASURF_TIMELINE, include/asurf.h); runs on the
instrumented build of the library (`make -C alphasurf_amd/csrc timeline` -> libasurf_hip_tl.so):
    python variants/timeline.py <kernel id> [bench.py flags]
prints the launch's span, the distribution of wave durations, the number of waves in flight over time and how the waves
spread over XCDs / CUs."""
import ctypes as C, json, os, sys
kid = int(sys.argv[1])
os.environ["ASURF_TIMELINE"] = str(kid)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ASURF_LIB"] = os.path.join(root, "alphasurf_amd", "csrc", "libasurf_hip_tl.so")
sys.path.insert(0, root)
import numpy as np
import bench
from alphasurf_amd import lib as _L

sys.argv = ["bench.py", "--no-cpu-baseline", "--no-also", "--steps", "3", "--warmup", "2"] + sys.argv[2:]
bench.main()
cap = 1 << 18
buf = np.zeros(cap * 3, dtype=np.uint64)
L = _L.lib()
L.asurf_debug_read_timeline.argtypes = [C.c_void_p, C.c_int]
L.asurf_debug_read_timeline.restype = C.c_int
n = L.asurf_debug_read_timeline(buf.ctypes.data, cap)
rec = buf.reshape(cap, 3)[:n]
t0, t1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
hw = (rec[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
xcc = ((rec[:, 2] >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
# only the records of the last launch: those that started within 5 ms of the latest start
last = t0.max()
m = (t1 > 0) & (t0 > last - 500000)
t0, t1, hw, xcc = t0[m], t1[m], hw[m], xcc[m]
start = t0.min()
us0, us1 = (t0 - start) / 100.0, (t1 - start) / 100.0
dur = us1 - us0
out = {"kernel": kid, "waves": int(m.sum()), "span_us": float(us1.max()),
       "wave_us": {k: round(float(np.percentile(dur, p)), 2) for k, p in
                   (("p10", 10), ("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))},
       "wave_us_mean": round(float(dur.mean()), 2),
       "sum_wave_us": round(float(dur.sum()), 1),
       "last_start_us": round(float(us0.max()), 2)}
# waves in flight, sampled every 1/20 of the span
ts = np.linspace(0, us1.max(), 21)
out["in_flight"] = [int(((us0 <= t) & (us1 > t)).sum()) for t in ts]
out["started_by"] = [int((us0 <= t).sum()) for t in ts]
# mean duration of the waves that started in each twentieth of the span, and the launch-order index they had
bins = np.minimum((us0 / max(us1.max(), 1e-9) * 20).astype(int), 19)
out["dur_by_start_bin"] = [round(float(dur[bins == b].mean()), 1) if (bins == b).any() else None for b in range(20)]
widx0 = np.nonzero(m)[0]
out["wave_index_by_start_bin"] = [int(np.median(widx0[bins == b])) if (bins == b).any() else None for b in range(20)]
cu = (hw >> 8) & 15
se = (hw >> 13) & 7
sh = (hw >> 12) & 1
simd = (hw >> 4) & 3
place = xcc * 1024 + se * 64 + sh * 32 + cu
uniq, cnt = np.unique(place, return_counts=True)
out["cus_used"] = int(uniq.size)
out["waves_per_cu"] = {"min": int(cnt.min()), "max": int(cnt.max()), "mean": round(float(cnt.mean()), 1)}
out["waves_per_xcc"] = [int((xcc == x).sum()) for x in range(8)]
# busy time per CU (sum of wave durations) to see imbalance
busy = np.zeros(uniq.size)
for i, u in enumerate(uniq):
    busy[i] = dur[place == u].sum()
out["busy_us_per_cu"] = {"min": round(float(busy.min()), 1), "max": round(float(busy.max()), 1),
                         "mean": round(float(busy.mean()), 1)}
# does block b run on XCD b % 8?  (wave index -> block: 4 waves per block for the list kernels, see the launch)
wpb = int(os.environ.get("TL_WAVES_PER_BLOCK", "4"))
widx = np.nonzero(m)[0]
out["block_mod8_is_xcc"] = round(float((((widx // wpb) % 8) == xcc).mean()), 4)
# the heaviest waves: where in the launch order are they?
idx = np.nonzero(m)[0]
top = np.argsort(-dur)[:10]
out["slowest"] = [{"wave": int(idx[i]), "us": round(float(dur[i]), 1), "start": round(float(us0[i]), 1)} for i in top]
print(json.dumps(out))
