#!/usr/bin/env python
"""Finds loads that hipcc waits for ONE BY ONE in the small kernels (static, no GPU):

  python profiles/tools/scan_serial_loads.py [vc_fused_kernels.hip ...] [--kernels SUBSTR,SUBSTR] [--min-loads 4] [--min-waits 3]

Round 6 (profiles/r06_hist_split.md): where a select, a float -> double conversion or a pointer dereference stands DIRECTLY behind a
load, hipcc tends to reuse one destination register and to put `s_waitcnt vmcnt(0)` behind every load of an unrolled sequence -- N
dependent memory round trips (0.2-0.3 us each behind a 400 MB stream) instead of one.  Found that way: the quarter blocks' table
prefetch (24 loads, 5.8 us), the dense histogram block of K_pre (K = 3 particles - 14 us per step), the loss block's long list, the
nu_omega prefetch, the slot pointers of the folded exchange.

The scan compiles a translation unit to gfx950 assembly, cuts it into kernels and reports every CLUSTER of vector-memory loads
(`global_load*` / `flat_load*`, fewer than 20 lines apart) that holds at least --min-loads loads and at least --min-waits
`s_waitcnt vmcnt(0)`: lines "kernel: [(first line, last line, loads, waits), ...]".  Legitimate dependent chains (a task record whose
fields steer the next load) show up too -- the list is short enough to read.  K_main's own loads are audited by check_asm_loads.py."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "velocycle_amd", "csrc")


def compile_asm(tu, out):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-falign-loops=64", "--cuda-device-only", "-S",
                    "-Wno-unused-variable", "-o", out, tu], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)


def scan(asm_path, min_loads=4, min_waits=3, gap=20):
    """{kernel symbol: [(first, last, loads, waits), ...]} for the clusters that qualify."""
    T = open(asm_path).read().splitlines()
    starts = [(i, l.split(":")[0]) for i, l in enumerate(T) if re.match(r"^_Z\w+:", l)]
    found = {}
    for n, (st, name) in enumerate(starts):
        en = starts[n + 1][0] if n + 1 < len(starts) else len(T)
        L = T[st:en]
        idx = [i for i, l in enumerate(L) if re.search(r"\b(global_load|flat_load)_dword", l)]
        out, i = [], 0
        while i < len(idx):
            j = i
            while j + 1 < len(idx) and idx[j + 1] - idx[j] < gap:
                j += 1
            if j - i + 1 >= min_loads:
                waits = sum(1 for l in L[idx[i]:idx[j] + 1] if "s_waitcnt vmcnt(0)" in l)
                if waits >= min_waits:
                    out.append((idx[i], idx[j], j - i + 1, waits))
            i = j + 1
        if out:
            found[name] = out
    return found


def demangled_hint(sym):
    """vc_tail2_kernelILi6ELi19EE... -> 'vc_tail2_kernel<6,19>' (enough to read the list)."""
    m = re.match(r"_Z\d+(\w+?)I(.*?)E[Ev]", sym)
    if not m:
        return sym
    args = re.findall(r"L[ib](\d+)E", m.group(2))
    return f"{m.group(1)}<{','.join(args)}>"


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = {a.split("=")[0]: (a.split("=")[1] if "=" in a else None) for a in sys.argv[1:] if a.startswith("--")}
    for k in list(opts):
        if opts[k] is None and k in ("--kernels", "--min-loads", "--min-waits"):
            raise SystemExit(f"{k}=VALUE")
    tus = args or ["vc_fused_kernels.hip", "vc_small_kernels.hip", "vc_generic_kernels.hip", "vc_p2p_exchange.hip"]
    want = [s for s in (opts.get("--kernels") or "").split(",") if s]
    total = 0
    for tu in tus:
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, "k.s")
            compile_asm(tu, asm)
            res = scan(asm, int(opts.get("--min-loads") or 4), int(opts.get("--min-waits") or 3))
        print(f"== {tu}: {len(res)} kernel(s) with clusters")
        for sym, cl in res.items():
            hint = demangled_hint(sym)
            if want and not any(w in hint for w in want):
                continue
            total += len(cl)
            print(f"  {hint}: {cl}")
    sys.exit(0)
