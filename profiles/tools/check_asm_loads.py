#!/usr/bin/env python
"""Static audit of the likelihood kernel's hand-placed count loads (VC_ASM_LOADS, csrc/vc_main_kernel.h).

An inline-asm `global_load_dwordx{2,4}` is invisible to hipcc's wait-count bookkeeping: the compiler believes the destination
tuple is written when the statement ends, so nothing but our own `s_waitcnt vmcnt(k)` statements keeps a consumer behind the
data, and nothing at all keeps a compiler-inserted copy / spill / re-use of those registers behind it
(/opt/skills/guides/cdna_hip_programming.md section 5.7, item 1).  This tool proves, on the assembly hipcc emits, that

  1. no instruction outside our asm statements reads or writes a register of a tuple while its load may still be in flight
     (between the load and the first of our waits that retires it) -- on the fall-through path of the kernel AND around the
     cell loop's back edge;
  2. the kernel uses no scratch (no spill could have parked a tuple) and stays within its launch bound's VGPR budget;
  3. every tuple is retired by a `vmcnt(0)` drain before the kernel's epilogue re-uses the registers;
  4. each wait inside the loop leaves exactly PF x (loads per fetch) operations outstanding.

In-flight model: our asm loads in program order; a wait `vmcnt(N)` retires all but the N youngest of them.  Stores and the
compiler's own loads also occupy vmcnt slots, in order, so the hardware retires AT LEAST what this model retires: the audit is
conservative (it can flag a safe program, never pass an unsafe one) as long as every asm load is followed by an asm wait on the
same path, which (3) checks.

  python profiles/tools/check_asm_loads.py [TU.hip ...]       (default: the instantiation families bench.py runs)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "velocycle_amd", "csrc")
DEFAULT_TUS = ["vc_main_vfull_nb_u16.hip", "vc_main_vu_nb_u16.hip", "vc_main_phase_nb_u16.hip"]
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LOAD = re.compile(r"^\s*global_load_dwordx(\d)\s+(v\[\d+:\d+\]|v\d+)\s*,")
WAIT = re.compile(r"^\s*s_waitcnt\s+vmcnt\((\d+)\)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)|^\s*s_branch\s+(\.LBB\d+_\d+)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def device_asm(tu, extra=()):
    out = os.path.join(tempfile.mkdtemp(), os.path.basename(tu).replace(".hip", ".s"))
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", *extra,
                    "-Wno-unused-variable", "-o", out, tu], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(asm_text):
    """{mangled name: (body lines, metadata dict)} of every vc_main_kernel instantiation in a device assembly file."""
    res, cur, name = {}, None, None
    for line in asm_text.splitlines():
        m = re.match(r"^(_Z14vc_main_kernel\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                res[name] = [cur, {}]
                cur = None
            else:
                cur.append(line)
    for name in res:
        m = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", asm_text, re.S)
        meta = {}
        if m:
            for k in ("private_segment_fixed_size", "next_free_vgpr", "accum_offset"):
                mm = re.search(r"\.amdhsa_" + k + r"\s+(\d+)", m.group(1))
                if mm:
                    meta[k] = int(mm.group(1))
        res[name][1] = meta
    return res


def audit(lines):
    """-> (violations, stats).  One linear pass over the kernel text, then a second pass over every loop that contains asm
    loads, entered with the in-flight state its back edge carries."""
    labels = {}
    for i, l in enumerate(lines):
        m = LABEL.match(l)
        if m:
            labels[m.group(1)] = i
    loops = []           # (header index, back-edge index)
    for i, l in enumerate(lines):
        m = BRANCH.match(l)
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt <= i:
                loops.append((tgt, i))
    violations, waits_in_loop, n_loads = [], [], 0

    def scan(a, b, inflight, record):
        """lines[a:b]; inflight = list of register sets, oldest first."""
        nonlocal n_loads
        in_asm = False
        for i in range(a, b):
            l = lines[i]
            if ";;#ASMSTART" in l:
                in_asm = True
                continue
            if ";;#ASMEND" in l:
                in_asm = False
                continue
            code = l.split(";")[0]
            if not code.strip() or code.strip().startswith(".") or LABEL.match(code):
                continue
            if in_asm:
                m = LOAD.match(code)
                if m:
                    inflight.append(regs_of(m.group(2)))
                    if record:
                        n_loads += 1
                    continue
                m = WAIT.match(code)
                if m:
                    n = int(m.group(1))
                    if record is not None and isinstance(record, list):
                        record.append((i, n, len(inflight)))
                    del inflight[: max(0, len(inflight) - n)]
                continue
            if inflight:
                hot = set().union(*inflight)
                bad = regs_of(code) & hot
                if bad:
                    violations.append((i, code.strip(), sorted(bad)))
        return inflight

    state = scan(0, len(lines), [], True)
    end_inflight = len(state)
    for hdr, back in loops:
        body = lines[hdr:back + 1]
        if not any(LOAD.match(x.split(";")[0]) for x in body):
            continue
        # state at the back edge = state after a pass from the kernel start to `back`
        st = scan(0, back + 1, [], False)
        rec = []
        scan(hdr, back + 1, st, rec)
        waits_in_loop = rec
    # the same violation can be found by both passes
    seen, uniq = set(), []
    for v in violations:
        if v[0] not in seen:
            seen.add(v[0])
            uniq.append(v)
    return uniq, {"asm_loads": n_loads, "in_flight_at_end": end_inflight, "loop_waits": waits_in_loop}


def report_tu(tu, extra=()):
    """One dict per vc_main_kernel instantiation of a translation unit: name, asm_loads, loop_waits, vgpr, scratch, problems.
    A kernel with scratch is not selectable (vc_finalize takes 8 genes per lane only when the code object has none, and
    VC_MAX_SCRATCH is ignored in asm-load builds): its problems are reported but do not fail the audit."""
    out = []
    text = device_asm(tu, extra)
    for name, (lines, meta) in sorted(kernels(text).items()):
        v, st = audit(lines)
        problems = []
        if st["asm_loads"] == 0:
            problems.append("no asm loads found (VC_ASM_LOADS=0 build?)")
        if v:
            problems.append(f"{len(v)} instruction(s) touch a tuple whose load may be in flight, first: line {v[0][0]}: {v[0][1]} {v[0][2]}")
        if st["in_flight_at_end"]:
            problems.append(f"{st['in_flight_at_end']} asm load(s) never retired by a drain")
        outs = sorted({n for _, n, _ in st["loop_waits"]})
        if len(outs) > 1:
            problems.append(f"loop waits leave different counts outstanding: {outs}")
        out.append({"name": name, "asm_loads": st["asm_loads"], "loop_waits": outs, "vgpr": meta.get("next_free_vgpr"),
                    "scratch": meta.get("private_segment_fixed_size", 0), "problems": problems})
    return out


def check_tu(tu, extra=(), verbose=True):
    ok = True
    for r in report_tu(tu, extra):
        bad = bool(r["problems"]) and r["scratch"] == 0
        tag = "FAIL" if bad else ("n/a " if r["scratch"] else "ok  ")
        if verbose or bad:
            print(f"{tag} {os.path.basename(tu)} {r['name']}: {r['asm_loads']} asm loads, loop waits vmcnt{r['loop_waits']}, vgpr {r['vgpr']}, "
                  f"scratch {r['scratch']}" + (" (not selectable)" if r["scratch"] else "") + ("" if not r["problems"] else " -- " + "; ".join(r["problems"])))
        ok = ok and not bad
    return ok


if __name__ == "__main__":
    tus = sys.argv[1:] or DEFAULT_TUS
    good = all([check_tu(t) for t in tus])
    sys.exit(0 if good else 1)
