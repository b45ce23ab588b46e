#!/usr/bin/env python
"""Rows of velocycle_amd/csrc/vc_tail_spec_rows.inc: the size-independent signatures (csrc/vc_common.h VcSig) of the workloads the
library compiles specialised small kernels for, read back from a finalized engine (vc_dbg_signature) -- not typed by hand.

  python profiles/tools/print_signature.py > velocycle_amd/csrc/vc_tail_spec_rows.inc        (on a GPU box; then rebuild)

A row: {name, kinds of launch (bits: 1 one-launch tail | 2 merged tail of the tutorial flow | 4 phases A / B of a sharded rank | 8 the K-particle step), MQ of the
gene blocks (2 | 4 | 6 | 14 >= the likelihood kernel's gene-level rows), {signature ints in VC_SIG_FIELDS order, cond}}."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from velocycle_amd.engine import HipEngine  # noqa: E402
from velocycle_amd.tuning import Tuning  # noqa: E402
from velocycle_amd.workloads import make_phase_spec, make_velocity_spec  # noqa: E402

dev = torch.device("cuda:0")
rows = []


def mq_of(nq):
    return 2 if nq <= 2 else (4 if nq <= 4 else (6 if nq <= 6 else 14))


FIELDS = ("model guide noise with_dnu kind H Nh Hw Nhw Nb Nx R NW K Kq nbk onehot pw_inline nq nco hist_has_S hist_has_U hist_dense hist_par "
          "nmat_r generic cond").split()
OPEN = ("Nb", "Nx", "NW", "K")           # what a "multi" row leaves open (+ pw_inline on a single rank: 4 | 8 floats per row follows NW)


def add(name, spec, world=1, multi=False):
    eng = HipEngine(spec, device=dev, rank=0, world_size=world, tuning=Tuning(no_tail_spec=True))
    sig = eng.signature()
    st = eng.stats
    nq = sig[FIELDS.index("nq")]
    if world > 1:
        kind = 4
    elif st["launches_per_step"] == 2 and "vu" in st["main_kernel"]:
        kind = 2
    elif st["launches_per_step"] == 2:
        kind = 1
    else:
        kind = 0
    if world == 1 and kind:
        kind |= 8            # the K-particle step of a single rank runs the same configuration
    eng.close()
    if kind == 0:
        print(f"// {name}: three launches per step on this configuration -- no row", file=sys.stderr)
        return
    if multi:
        assert sig[FIELDS.index("onehot")] == 1, name
        for f in OPEN + (("pw_inline",) if world == 1 and sig[FIELDS.index("pw_inline")] != 0 else ()):
            sig[FIELDS.index(f)] = -1
    for r in rows:
        if r[3] == sig:          # the same signature under another launch structure: one row, both kinds
            r[1] |= kind
            r[0] = min(r[0], name, key=len)
            print(f"// {name}: same signature as row {r[0]} (kinds now {r[1]})", file=sys.stderr)
            return
    rows.append([name, kind, mq_of(nq), sig])


NC, NG = 50000, 2000
# ranks of a sharded run first: a single-rank "multi" row leaves pw_inline open and would otherwise shadow its rank row
# (first match wins: the CLOSED rows of a configuration -- one sample, exactly two samples -- stand in front of its open "multi" row)
for world in (8, 1):
    tag = "_rank" if world > 1 else ""
    # velocity: mean-field / LRMN (the reference's default model_type) guide x nothing conditioned / the tutorials' conditioning
    for name, mode in (("vjoint", "vjoint"), ("vcond", "vcond"), ("vjoint_lrmn", "vjoint_lrmn"), ("vcond_mf", "vcond_mf")):
        add(name + tag, make_velocity_spec(NC, NG, mode, 1, 1, seed=0, device=dev), world)
    # round 6: the tutorials' FIRST velocity stage -- constant angular speed, AngularSpeed.trivial_prior(harmonics=0) /
    # omega_n_harmonics=0 (Tutorial_Capolupo_HumanFibroblasts_OneSample.ipynb:690,721; angularspeed.py:311-354)
    add("vcond_hw0" + tag, make_velocity_spec(NC, NG, "vcond", 1, 0, seed=0, device=dev), world)
    add("phase" + tag, make_phase_spec(NC, NG, seed=0, device=dev), world)
    # round 6: the default of preprocess_for_phase_estimation / preprocess_for_velocity_estimation -- n_harmonics = 2
    # (preprocessing.py:108,217; the package tutorials pass 1)
    for name, mode in (("vjoint_h2", "vjoint"), ("vcond_h2", "vcond")):
        add(name + tag, make_velocity_spec(NC, NG, mode, 1, 1, seed=0, device=dev, H=2), world)
    add("phase_h2" + tag, make_phase_spec(NC, NG, seed=0, device=dev, H=2), world)
    # round 6: exactly two samples (BASELINE configs[4], Tutorial_Aissa_PC9_TwoSample: Nx = Nb = 2) with every count closed
    for name, mode, hw in (("vjoint_2s", "vjoint", 1), ("vcond_2s", "vcond", 1), ("vcond_hw0_2s", "vcond", 0)):
        add(name + tag, make_velocity_spec(NC // 2, NG, mode, 2, hw, seed=0, device=dev), world)
    for name, mode in (("vjoint", "vjoint"), ("vcond", "vcond"), ("vjoint_lrmn", "vjoint_lrmn"), ("vcond_mf", "vcond_mf")):
        add(name + "_multi" + tag, make_velocity_spec(NC // 2, NG, mode, 2, 1, seed=0, device=dev), world, multi=True)
    add("vcond_hw0_multi" + tag, make_velocity_spec(NC // 2, NG, "vcond", 2, 0, seed=0, device=dev), world, multi=True)
    add("phase_multi" + tag, make_phase_spec(NC // 2, NG, seed=0, device=dev, n_batches=2), world, multi=True)
print("// rows of VC_SPECS (vc_tail_spec.h): {name, kinds of launch, MQ of the gene blocks, signature} -- printed by profiles/tools/print_signature.py")
print("// signature = " + " ".join(FIELDS) + "; -1 = left open")
for name, kind, mq, sig in rows:
    print('  {"%s", %d, %d, {%s, %du}},' % (name, kind, mq, ", ".join(str(x) for x in sig[:-1]), sig[-1] & 0xFFFFFFFF))
