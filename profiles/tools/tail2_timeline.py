#!/usr/bin/env python
"""Per-wave wall-clock timeline of the step's second launch (vc_tail2_kernel, or K_tail + K_omega with VC_TAIL2=0), from the
VC_DBG_TIMES build:
  make -C velocycle_amd/csrc BUILD=build_dbg OUT=../../scratch/libs/dbg.so EXTRA=-DVC_DBG_TIMES
  VC_LIB_PATH=$PWD/scratch/libs/dbg.so python profiles/tools/tail2_timeline.py [vjoint|phase|vcond] [cells] [genes] [samples]
Stamps are s_memrealtime (100 MHz): 1 tick = 10 ns.  Printed: K_main's span, the gap to the first wave of the next launch, when
each stage of the gene / cell blocks is reached (relative to that launch's first wave entry; median / max over waves), the
stages of the nu_omega chain inside the cell blocks, the last stamp of the launch, and the gap back to K_main."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from velocycle_amd.engine import HipEngine
from velocycle_amd.tuning import Tuning
from velocycle_amd.svi import SVIRunner
from velocycle_amd.workloads import make_phase_spec, make_velocity_spec

mode = sys.argv[1] if len(sys.argv) > 1 else "vjoint"
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
NG = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
dev = torch.device("cuda:0")
NCOND = int(sys.argv[4]) if len(sys.argv) > 4 else 1       # samples (conditions = batches): NC cells in total
spec = (make_phase_spec(NC // NCOND, NG, seed=0, device=dev, n_batches=NCOND) if mode == "phase"
        else make_velocity_spec(NC // NCOND, NG, mode, NCOND, 1, seed=0, device=dev))
eng = HipEngine(spec, device=dev, tuning=Tuning.from_env())
nwg = eng.stats["main_grid"]
print(mode, NC, NG, eng.stats["main_kernel"], "launches per step", eng.stats["launches_per_step"], "pw_inline", eng.stats["pw_inline"])
run = SVIRunner(eng, {"lr": 0.03, "lrd": 0.999, "betas": (0.8, 0.99)}, mode="perf", seed=0, use_graph=False)
run.run_perf(60, sync=True)
del run
eng.dump_dbg_times("/tmp/vc_times.bin")      # (a -DVC_DBG_TIMES build of the library: VC_LIB_PATH)
eng.close()
del eng
raw = np.fromfile("/tmp/vc_times.bin", dtype=np.uint64).astype(np.int64)
main = raw[: nwg * 32].reshape(-1, 8)
w = raw[nwg * 32 + 3 * 4096 * 8:].reshape(2, 4096, 16, 8)
mt0 = main[main[:, 0] > 0][:, 0].min()
mt3 = main[:, 3].max()
print(f"K_main (last launch of the run): first entry -> last exit {(mt3 - mt0) / 100:.2f} us; median wave exit {(np.median(main[:, 3]) - mt0) / 100:.2f}")
NGB = ((NG + 63) // 64 * 64 + 255) // 256 * 256 // 64 if False else None
s0 = w[0]
live = s0[:, :, 0] > 0
# (blocks that lead the grid as histogram blocks may hold kid-0 stamps of an EARLIER launch structure of the same run: the boot launches)
# (round 6: a histogram block of a gene block whose quarter blocks do its work leaves without a stamp -- the lead ends behind the LAST
# stamped histogram block, plus any such unstamped block behind it)
_h7 = np.where((w[1][:, :, 7] > 0).any(1))[0]
_hist_lead = int(_h7.max()) + 1 if len(_h7) else 0
live[:_hist_lead] = False
while _hist_lead < w.shape[1] and len(_h7) and not live[_hist_lead].any():
    _hist_lead += 1
z = s0[:, :, 0][live].min()
print(f"K_main last exit -> first wave of the next launch: {(z - mt3) / 100:.2f} us")
nblk = np.where(live.any(1))[0]
print(f"blocks with stamps: {len(nblk)} (indices {nblk.min()}..{nblk.max()})")
u = (s0 - z) / 100.0
ngb = int(os.environ.get("NGB", 0)) or (((NG + 511) // 512) * 512 // 64)


def show(tag, arr, sel, ks=range(8)):
    for k in ks:
        v = arr[:, :, k][sel & (arr[:, :, k] > -1e8)]
        if len(v):
            print(f"   {tag:34s} stamp{k}: min {v.min():6.2f} med {np.median(v):6.2f} max {v.max():6.2f}  (n={len(v)})")


u[s0 <= 0] = -1e9
# (round 5: the histogram blocks lead the grid of the one-launch tail -- the gene blocks start behind them)
lead = _hist_lead
gene = np.zeros_like(live); gene[lead:lead + ngb] = True
show("gene blocks (all waves)", u, live & gene)
for wv, role in ((0, "nu[0]"), (12, "shape_inv"), (13, "role 14"), (15, "role 13")):
    r = np.zeros_like(live); r[lead:lead + ngb, wv] = True
    show(f"gene wave {wv} ({role})", u, live & r)
cell = np.zeros_like(live); cell[lead + ngb:] = True
show("cell blocks", u, live & cell)
s1 = w[1]
live1 = s1[:, :, 1] > 0
if live1.any():
    u1 = (s1 - z) / 100.0
    u1[s1 <= 0] = -1e9
    show("nu_omega chain (same clock origin)", u1, live1, ks=range(1, 6))
    end = max(u[u > -1e8].max(), u1[u1 > -1e8].max())
else:
    end = u[u > -1e8].max()
for k, tag in ((6, "histogram waves: update re-derived"), (7, "histogram waves: done"), (0, "loss block: done")):
    sel = s1[:, :, k] > 0
    if sel.any():
        v = (s1[:, :, k][sel] - z) / 100.0
        print(f"   {tag:34s} stamp{k}: min {v.min():6.2f} med {np.median(v):6.2f} max {v.max():6.2f}  (n={len(v)})")
        end = max(end, v.max())
print(f"   last stamp of the launch at {end:.2f} us")
# the slowest histogram blocks, one line each: block index, when its update was re-derived / when it was done (max over its waves)
sel = s1[:, :, 7] > 0
if sel.any():
    done = np.where(sel, (s1[:, :, 7] - z) / 100.0, -1e9).max(1)
    red = np.where(s1[:, :, 6] > 0, (s1[:, :, 6] - z) / 100.0, -1e9).max(1)
    first = np.where(s1[:, :, 6] > 0, (s1[:, :, 6] - z) / 100.0, 1e9).min(1)
    order = np.argsort(-done)[:6]
    print("   slowest histogram blocks (block: first wave re-derived / last wave re-derived / done): " +
          "; ".join(f"{int(i)}: {first[i]:.2f} / {red[i]:.2f} / {done[i]:.2f}" for i in order))
    blk = np.where(sel.any(1))[0]
    print(f"   histogram blocks {blk.min()}..{blk.max()}: done by block index (every 8th): " +
          " ".join(f"{done[i]:.1f}" for i in blk[::8]))
if os.environ.get("VC_TL_HISTDETAIL"):
    for i in order:
        print(f"   hist block {int(i)} waves 0..3 stamps 2 (chunks added), 4 (table rows requested, snapshot and sums read), 6 (update re-derived), 7 (done):", [[round(float((s1[i, wv, k] - z) / 100.0), 2) for k in (2, 4, 6, 7)] for wv in range(4)])
    for i in (0, 1):
        print(f"   hist block {int(i)} waves 0..3 stamps 2 (chunks added), 4 (table rows requested, snapshot and sums read), 6 (update re-derived), 7 (done):", [[round(float((s1[i, wv, k] - z) / 100.0), 2) for k in (2, 4, 6, 7)] for wv in range(4)])
if os.environ.get("VC_TL_HISTDETAIL"):
    e3 = s1[:, :, 3]
    blk3 = np.where((e3 > 0).any(1) & ~(s1[:, :, 1] > 0).any(1))[0]      # eps blocks: stamp 3 without the chain's stamp 1
    if len(blk3):
        v = (e3[blk3][e3[blk3] > 0] - z) / 100.0
        print(f"   eps blocks {blk3.min()}..{blk3.max()}: end min {v.min():.2f} med {np.median(v):.2f} max {v.max():.2f}")
