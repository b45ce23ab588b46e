#!/usr/bin/env python
"""Static VALU count of the likelihood kernel's cell loop, from the device assembly hipcc emits.

For every requested instantiation of vc_main_kernel the main loop (the largest innermost loop of the kernel: two or three
cells per trip, the rotating register buffers) is located in `hipcc --cuda-device-only -S` output and its VALU instructions
are counted: all `v_*` (one issue slot per wave64 instruction; packed-math ops retire two genes per slot) and, among
them, the transcendentals (v_exp/v_log/v_rcp/v_rsq/v_sqrt/v_sin/v_cos: quarter rate).  profiles/tools/valu_rate.hip
measures on the GPU what a SIMD needs for exactly that instruction MIX and nothing else (no loads, no cross-lane work,
16 independent chains) at 1..8 waves per SIMD; its output (profiles/r03_valu_rate.txt) is folded in here, scaled by
the instruction count of each instantiation: the arithmetic floor of the cell loop that bench.py reports next to the
HBM roofline (`roofline.valu`).  The counting runs without a GPU.

  python profiles/tools/valu_count.py            -> writes profiles/valu_model.json
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "velocycle_amd", "csrc")
TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_")
TRANS_SLOW = re.compile(r"^v_(exp|log|sin|cos)_")
# (translation unit, H, NB, KIND, NOISE, GPL, C16) -- the instantiations bench.py runs, float32 and uint16 count storage
KERNELS = [("vc_main_vfull_nb_u16.hip", 1, 0, 1, 0, 8, 1), ("vc_main_vu_nb_u16.hip", 1, 0, 2, 0, 8, 1),
           ("vc_main_phase_nb_u16.hip", 1, 0, 0, 0, 8, 1),
           ("vc_main_vfull_nb.hip", 1, 0, 1, 0, 8, 0), ("vc_main_vu_nb.hip", 1, 0, 2, 0, 8, 0), ("vc_main_phase_nb.hip", 1, 0, 0, 0, 8, 0),
           ("vc_main_vfull_nb_u16.hip", 1, 2, 1, 0, 8, 1), ("vc_main_vu_nb_u16.hip", 1, 2, 2, 0, 8, 1),
           # round 6: the U-only kernel with the nu_omega partials per lane (one condition, D == 1: every one-sample tutorial flow)
           ("vc_main_vu_nb_u16_pwl.hip", 1, 0, 2, 0, 8, 5), ("vc_main_vu_nb_pwl.hip", 1, 0, 2, 0, 8, 4)]
KIND_NAME = {0: "phase", 1: "vfull", 2: "vu"}
NOISE_NAME = {0: "nb", 1: "poisson", 2: "lognormal"}
MIX_OF_KIND = {0: "mix phase (S only)", 1: "mix vfull (S+U)", 2: "mix vu (U only)"}


def mix_table(path=os.path.join(ROOT, "profiles", "r03_valu_rate.txt")):
    """{mix name: {"instr": instructions per cell iteration of the mix, "ns": {waves per SIMD: ns per cell iteration}}},
    and the shader clock of the run (ticks per ns of the one-wave rows, where the stamping wave is the only one)."""
    mixes, clocks = {}, []
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        r = json.loads(line)
        if r["waves_per_simd"] == 1 and r["what"].startswith("mix"):
            clocks.append(r["ticks_per_ns"])
        if "ns_per_cell_iter" in r:
            m = mixes.setdefault(r["what"], {"instr": r["instr_per_cell_iter"], "ns": {}})
            m["ns"][str(r["waves_per_simd"])] = r["ns_per_cell_iter"]
    return mixes, round(sorted(clocks)[len(clocks) // 2], 3)


MIXES, MIX_CLOCK_GHZ = mix_table()


def device_asm(tu, cache={}):
    if tu not in cache:
        out = os.path.join(tempfile.mkdtemp(), tu.replace(".hip", ".s"))
        extra = os.environ.get("EXTRA", "").split()
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-falign-loops=64", "--cuda-device-only", "-S",
                        *extra, os.path.join(CSRC, tu), "-o", out], check=True, stderr=subprocess.DEVNULL)
        cache[tu] = open(out).read().splitlines()
    return cache[tu]


def kernel_body(lines, sym):
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start:end + 1]


def main_loop(body):
    """(first, last) line index of the largest loop: a label and the last backward branch to it."""
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    return best


def count(tu, H, NB, KIND, NOISE, GPL, C16):
    sym = f"_Z14vc_main_kernelILi{H}ELi{NB}ELi{KIND}ELi{NOISE}ELi{GPL}ELi{C16}EEv6VcDims6VcBufs"
    body = kernel_body(device_asm(tu), sym)
    a, b = main_loop(body)
    # basic blocks of the loop; the blocks that flush staged per-cell sums run once per 16 cells (LDS tile: they hold the
    # ds_read_b128 column sums) or once per 64 cells (DPP staging: a global_store inside the loop) and are weighted so
    blocks, cur = [], []
    for l in body[a:b + 1]:
        if l.startswith(".LBB") and cur:
            blocks.append(cur); cur = []
        cur.append(l)
        if re.match(r"\s+s_c?branch", l):
            blocks.append(cur); cur = []
    if cur:
        blocks.append(cur)
    ops, weights = [], []
    for blk in blocks:
        ins = [l.split()[0] for l in blk if l.startswith("\t") and not l.strip().startswith((";", "."))]
        w = 1.0
        if any(o.startswith("ds_read_b128") for o in ins):
            w = 1.0 / 16
        elif any(o.startswith("global_store") for o in ins):
            w = 1.0 / 64
        ops += ins
        weights += [w] * len(ins)
    wsum = lambda pred: sum(w for o, w in zip(ops, weights) if pred(o))
    isv = lambda o: o.startswith("v_")
    n_valu = wsum(isv)
    n_trans = wsum(lambda o: isv(o) and bool(TRANS.match(o)))
    n_slow = wsum(lambda o: isv(o) and bool(TRANS_SLOW.match(o)))
    n_pk = wsum(lambda o: o.startswith("v_pk_"))
    n_plain = n_valu - n_trans - n_pk
    # cells per loop trip = NBUF = prefetch depth + 1: the asm-load path has one hand-placed `s_waitcnt vmcnt(k)` per cell
    # (k > 0: the younger fetches stay in flight); round 2's compiler-counted loop had two cells per trip
    asm_waits = sum(1 for i, l in enumerate(body[a:b + 1]) if re.match(r"\s+s_waitcnt vmcnt\([1-9]\d*\)\s*$", l)
                    and ";;#ASMSTART" in body[a + i - 1])
    cells = asm_waits if asm_waits else 2
    res = {"valu_per_cell_iter": round(n_valu / cells, 2), "trans_per_cell_iter": round(n_trans / cells, 2),
           "packed_per_cell_iter": round(n_pk / cells, 2), "plain_per_cell_iter": round(n_plain / cells, 2),
           "exp_log_per_cell_iter": round(n_slow / cells, 2), "rcp_per_cell_iter": round((n_trans - n_slow) / cells, 2),
           "vmem_loads_per_cell_iter": round(wsum(lambda o: o.startswith(("global_load", "buffer_load"))) / cells, 2),
           "lds_per_cell_iter": round(wsum(lambda o: o.startswith("ds_")) / cells, 2),
           "salu_per_cell_iter": round(wsum(lambda o: o.startswith("s_")) / cells, 2),
           "genes_per_lane": GPL}
    mix = MIXES[MIX_OF_KIND[KIND]]
    res["mix"] = MIX_OF_KIND[KIND]
    res["floor_ns_per_cell_iter"] = {w: round(ns * res["valu_per_cell_iter"] / mix["instr"], 1) for w, ns in mix["ns"].items()}
    return f"vc_main_kernel<{H},{NB},{KIND_NAME[KIND]}_{NOISE_NAME[NOISE]},gpl{GPL}{',u16' if C16 & 1 else ''}{',pwl' if C16 & 4 else ''}>", res


def main():
    out = {"mix_clock_ghz": MIX_CLOCK_GHZ, "mixes": MIXES,
           "note": "static counts of the cell loop in the gfx950 assembly (profiles/tools/valu_count.py); "
                   "floor_ns_per_cell_iter[w] = what a SIMD needs at w waves for the kernel's instruction mix alone, measured "
                   "on MI355X (profiles/tools/valu_rate.hip -> profiles/r03_valu_rate.txt, at mix_clock_ghz) and scaled by "
                   "valu_per_cell_iter / the mix's instruction count; arithmetic floor of a launch = floor_ns_per_cell_iter"
                   "[waves per SIMD of the launch] x (gene blocks x cells) / (CUs x 4 SIMDs) x mix_clock / clock",
           "kernels": {}}
    for k in KERNELS:
        name, res = count(*k)
        out["kernels"][name] = res
        print(name, res)
    path = os.environ.get("VALU_MODEL_OUT", os.path.join(ROOT, "profiles", "valu_model.json"))
    try:        # measured on the GPU with the stamped build, not derivable here: carried over
        old = json.load(open(path))
        out.update({k: v for k, v in old.items() if k.startswith("in_loop_clock")})
    except Exception:
        pass
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
