#!/usr/bin/env python
"""Per-wave wall-clock timeline of the fused step's two small kernels (K_tail, K_omega), from the VC_DBG_TIMES build:
  make -C velocycle_amd/csrc BUILD=build_dbg OUT=../../scratch/libs/dbg.so EXTRA=-DVC_DBG_TIMES
  VC_LIB_PATH=$PWD/scratch/libs/dbg.so python profiles/tools/fused_timeline.py [mode] [cells]
Stamps are s_memrealtime (100 MHz): 1 tick = 10 ns.  Printed per kernel: when each phase is reached (relative to the
kernel's first wave entry), median / max over waves, split by block kind and role."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from velocycle_amd.engine import HipEngine
from velocycle_amd.tuning import Tuning
from velocycle_amd.svi import SVIRunner
from velocycle_amd.workloads import make_velocity_spec

mode = sys.argv[1] if len(sys.argv) > 1 else "vjoint"
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 6250
dev = torch.device("cuda:0")
spec = make_velocity_spec(NC, 2000, mode, 1, 1, seed=0, device=dev)
eng = HipEngine(spec, device=dev, tuning=Tuning.from_env())
nwg = eng.stats["main_grid"]
run = SVIRunner(eng, {"lr": 0.03, "lrd": 0.999, "betas": (0.8, 0.99)}, mode="perf", seed=0, use_graph=False)
run.run_perf(30, sync=True)
ng_blocks = (2000 + 63) // 64 if eng.stats["main_kernel"].endswith("gpl8>") or True else 0
del run
eng.dump_dbg_times("/tmp/vc_times.bin")      # (a -DVC_DBG_TIMES build of the library: VC_LIB_PATH)
eng.close()
del eng
raw = np.fromfile("/tmp/vc_times.bin", dtype=np.uint64).astype(np.int64)
main = raw[: nwg * 32].reshape(-1, 8)
w = raw[nwg * 32 + 3 * 4096 * 8:].reshape(2, 4096, 16, 8)
mt0 = main[main[:, 0] > 0][:, 0].min()
mt3 = main[:, 3].max()
print(f"{mode} {NC}: K_main first entry -> last exit {(mt3 - mt0) / 100:.2f} us")
NGB = 2048 // 64          # gene blocks of K_tail (Ng_pad / 64)
for kid, name in ((0, "K_tail"), (1, "K_omega")):
    s = w[kid]
    live = s[:, :, 0] > 0
    z = s[:, :, 0][live].min()
    print(f"{name}: K_main exit -> first entry {(z - mt3) / 100:.2f} us" if kid == 0 else
          f"{name}: K_tail last stamp -> first entry {(z - w[0][:, :, 6][w[0][:, :, 6] > 0].max()) / 100:.2f} us")
    u = (s - z) / 100.0
    def show(tag, sel):
        for k in range(8):
            v = u[:, :, k][sel & (s[:, :, k] > 0)]
            if len(v):
                print(f"   {tag:30s} stamp{k}: min {v.min():6.2f} med {np.median(v):6.2f} max {v.max():6.2f}  (n={len(v)})")
    if kid == 0:
        gene = np.zeros_like(live); gene[:NGB] = True
        show("gene blocks (all waves)", live & gene)
        # waves 12..15 hold the roles 12 (shape_inv), 14, 15 (cov_factor halves / mean-field log beta on 14), 13 (log gamma or LRMN core)
        for wv, role in ((0, "nu[0]"), (12, "shape_inv"), (13, "role 14"), (15, "role 13")):
            r = np.zeros_like(live); r[:NGB, wv] = True
            show(f"gene wave {wv} ({role})", live & r)
        cell = np.zeros_like(live); cell[NGB:] = True
        show("cell blocks", live & cell)
    else:
        b0 = np.zeros_like(live); b0[0] = True
        show("block 0", live & b0)
        rest = np.zeros_like(live); rest[1:] = True
        show("other blocks", live & rest)
    end = u[:, :, :][s > 0].max()
    print(f"   last stamp of the kernel at {end:.2f} us")
