"""Per-wave timeline of K_main (build the library with EXTRA=-DVC_DBG_TIMES first):
  make -C velocycle_amd/csrc clean; make -C velocycle_amd/csrc -j EXTRA=-DVC_DBG_TIMES
  python profiles/tools/wave_timeline.py vjoint 4 [cells]
Every wave stamps the constant-rate clock at entry, after the per-gene latents are loaded, at the end of its cell loop and
after the epilogue, plus HW_ID / XCC_ID; this prints the distributions and their breakdown by XCD, CU and dispatch order.
(The tick is nominally 10 ns; on the boxes used it ran ~16 % fast against rocprofv3 durations.)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from velocycle_amd.engine import HipEngine
from velocycle_amd.tuning import Tuning
from velocycle_amd.svi import SVIRunner
from velocycle_amd.workloads import make_velocity_spec
mode = sys.argv[1] if len(sys.argv) > 1 else "vjoint"
dev = torch.device("cuda:0")
NC = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
spec = make_velocity_spec(NC, 2000, mode, 1, 1, seed=0, device=dev)
eng = HipEngine(spec, device=dev, tuning=Tuning.from_env())
run = SVIRunner(eng, {"lr": 0.03, "lrd": 0.999, "betas": (0.8, 0.99)}, mode="perf", seed=0, use_graph=False)
run.run_perf(int(os.environ.get("VC_TIMELINE_STEPS", "600")), sync=True)      # long enough for the clocks to settle (the first ~30 ms run slower)
grid = int(eng.stats["main_grid"])
print("kernel", eng.stats["main_kernel"], "grid", grid)
del run
eng.dump_dbg_times("/tmp/vc_times.bin")      # (a -DVC_DBG_TIMES build of the library: VC_LIB_PATH)
eng.close()
del eng
import gc; gc.collect()
raw = np.fromfile("/tmp/vc_times.bin", dtype=np.uint64)[: grid * 32].reshape(-1, 8).astype(np.int64)   # the small kernels' stamps follow
raw = raw[raw[:, 0] > 0]
t = raw[:, :4]
z = t[:, 0].min()
us = (t - z) / 100.0        # 100 MHz
hw, xcc = raw[:, 4], raw[:, 5] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3
print("waves", len(us), "kernel span %.1f us" % us[:, 3].max())
for k, nm in enumerate(["entry", "latents loaded", "loop end", "epilogue end"]):
    v = us[:, k]
    print(f"{nm:16s} min {v.min():7.2f} p10 {np.percentile(v,10):7.2f} med {np.median(v):7.2f} p90 {np.percentile(v,90):7.2f} max {v.max():7.2f}")
d = us[:, 2] - us[:, 1]
if raw[:, 6].max() > 0:      # shader-clock ticks of the cell loop / its duration on the constant-rate clock (nominally 10 ns per tick)
    clk = raw[:, 6] / np.maximum(t[:, 2] - t[:, 1], 1) / 10.0
    sel = d > 0.5 * np.median(d)
    print("shader clock inside the cell loop (GHz, against the nominal 100 MHz constant clock): med %.3f p10 %.3f p90 %.3f" %
          (np.median(clk[sel]), np.percentile(clk[sel], 10), np.percentile(clk[sel], 90)))
print("loop duration    min %.2f med %.2f p90 %.2f max %.2f" % (d.min(), np.median(d), np.percentile(d, 90), d.max()))
if raw[:, 6].max() > 0:      # the same loop in SHADER cycles (what a removed stall shows up in, whatever clock the chip then holds)
    cyc = raw[:, 6].astype(np.float64)
    print("cell loop shader cycles per wave: med %.0f p10 %.0f p90 %.0f sum over waves %.4g" %
          (np.median(cyc), np.percentile(cyc, 10), np.percentile(cyc, 90), cyc.sum()))
wg = np.arange(len(us)) // 4
nGB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
print("by XCC_ID:", {int(x): round(float(np.median(d[xcc == x])), 1) for x in np.unique(xcc)})
print("by blockIdx%8:", {int(x): round(float(np.median(d[wg % 8 == x])), 1) for x in range(8)})
print("by gene block:", {int(x): round(float(np.median(d[wg % nGB == x])), 1) for x in range(nGB)})
print("by SE:", {int(x): round(float(np.median(d[se == x])), 1) for x in np.unique(se)})
print("by CU id:", {int(x): round(float(np.median(d[cu == x])), 1) for x in np.unique(cu)})
ch = wg // nGB
q = np.percentile(ch, [25, 50, 75])
print("by chunk quartile:", [round(float(np.median(d[(ch >= lo) & (ch < hi)])), 1) for lo, hi in [(0, q[0]), (q[0], q[1]), (q[1], q[2]), (q[2], 1e9)]])
# waves per physical CU
key = xcc * 10000 + se * 1000 + sh * 100 + cu
u, cnt = np.unique(key, return_counts=True)
print("waves per physical CU: ", dict(zip(*np.unique(cnt, return_counts=True))), "CUs used", len(u))
for c in np.unique(cnt):
    sel = np.isin(key, u[cnt == c])
    print("  CUs with %d waves: median loop %.1f" % (c, np.median(d[sel])))
