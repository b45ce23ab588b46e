import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from velocycle_amd.engine import HipEngine
from velocycle_amd.tuning import Tuning
from velocycle_amd.svi import SVIRunner
from velocycle_amd.workloads import make_velocity_spec
mode = sys.argv[1] if len(sys.argv) > 1 else "vcond"
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
dev = torch.device("cuda:0")
spec = make_velocity_spec(NC, 2000, mode, 1, 1, seed=0, device=dev)
eng = HipEngine(spec, device=dev, tuning=Tuning.from_env())
st = eng.stats
run = SVIRunner(eng, {"lr": 0.03, "lrd": 0.999, "betas": (0.8, 0.99)}, mode="perf", seed=0, use_graph=False)
run.run_perf(20, sync=True)
nwg = st["main_grid"]
eng.dump_dbg_times("/tmp/vc_times.bin")      # (a -DVC_DBG_TIMES build of the library: VC_LIB_PATH)
del run; eng.close(); del eng
raw = np.fromfile("/tmp/vc_times.bin", dtype=np.uint64).astype(np.int64)
main = raw[: nwg * 32].reshape(-1, 8)
small = raw[nwg * 32:].reshape(3, 4096, 8)
mt0 = main[main[:, 0] > 0][:, 0].min(); mt3 = main[:, 3].max()
print(f"{mode} {NC}: K_main span {(mt3 - mt0)/100:.2f} (ticks/100; ~16% long vs us)")
names = ["K_pre", "K_post", "K_fin+Adam"]
for kid in range(3):
    s = small[kid]; s = s[s[:, 0] > 0]
    if len(s) == 0: continue
    z = s[:, 0].min()
    u = (s - z) / 100.0
    end = u[:, 3].max()
    print(f"{names[kid]}: blocks {len(s)}, span {end:.2f}; entry med {np.median(u[:,0]):.2f} max {u[:,0].max():.2f}")
    for k in (1, 2, 3):
        v = u[:, k][s[:, k] > 0]
        if len(v): print(f"   stamp{k}: min {v.min():.2f} med {np.median(v):.2f} max {v.max():.2f} (n={len(v)})")
    if kid == 1:
        ng = (2000 + 63) // 64 if True else 0
        g = u[:32]; c = u[32:]
        print(f"   gene blocks: GO loop done med {np.median(g[:,1]):.2f}, roles done med {np.median(g[:,2]):.2f}, end med {np.median(g[:,3]):.2f} max {g[:,3].max():.2f}")
        if len(c): print(f"   cell blocks: loads done med {np.median(c[:,1]):.2f}, end med {np.median(c[:,3]):.2f} max {c[:,3].max():.2f}")
    if kid == 0:
        print(f"   block 0..7 (gene) end: {np.round(u[:8,3],2)}; last blocks end: {np.round(u[-4:,3],2)}")
    if kid == 2:
        print(f"   block 0: fin done {u[0,1]:.2f}, adam done {u[0,2]:.2f}, end {u[0,3]:.2f}; other blocks end med {np.median(u[1:,3]):.2f} max {u[1:,3].max():.2f}")
    # gap between kernels
print("gaps: pre end -> main start:", (mt0 - small[0][small[0][:,3]>0][:,3].max())/100, " main end -> post start:", (small[1][small[1][:,0]>0][:,0].min() - mt3)/100,
      " post end -> fin start:", (small[2][small[2][:,0]>0][:,0].min() - small[1][small[1][:,3]>0][:,3].max())/100)
