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
DEFAULT_TUS = ["vc_main_vfull_nb_u16.hip", "vc_main_vu_nb_u16.hip", "vc_main_phase_nb_u16.hip", "vc_main_vu_nb_u16_noloss.hip",
               "vc_main_vfull_nb_u16_noloss.hip", "vc_main_phase_nb_u16_noloss.hip", "vc_main_vu_nb_u16_pwl.hip", "vc_main_vu_nb_u16_pwl_noloss.hip"]
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
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-falign-loops=64", "--cuda-device-only", "-S", *extra,
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


SMOV = re.compile(r"^\s*s_mov_b64\s+s\[(\d+):(\d+)\],\s*(-1|0)\s*$")
SCSEL = re.compile(r"^\s*s_cselect_b64\s+s\[(\d+):(\d+)\],\s*(-1,\s*0|0,\s*-1)\s*$")
VCC_AND = re.compile(r"^\s*s_(and|andn2)_b64\s+vcc,\s*exec,\s*s\[(\d+):(\d+)\]\s*$")
VCC_BR = re.compile(r"^\s*s_cbranch_vcc(nz|z)\s+(\.LBB\d+_\d+)")
SDST = re.compile(r"^\s*[sv]_\w+\s+(s\d+|s\[\d+:\d+\]|vcc(?:_lo|_hi)?)\b")
EXIT_MARK = "vc_loop_exit"


def sregs_of(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(lines):
    """-> (violations, stats).  Forward data-flow over the kernel's control-flow graph.  The state carried along a path:

      * the QUEUE of asm-load tuples that may still be in flight (oldest first): an asm load appends its tuple, our
        `s_waitcnt vmcnt(N)` retires all but the N youngest;
      * the SGPR pairs that hold a wave-uniform boolean (-1 / 0): `s_mov_b64 s[a:b], -1|0` (known), `s_cselect_b64 s[a:b],
        -1, 0` (unknown, but the same value wherever it is tested): `s_and[n2]_b64 vcc, exec, s[a:b]` + `s_cbranch_vcc[n]z`
        either follows the known value or splits the path and pins the boolean on each side.  hipcc turns an exit from the
        unrolled cell loop into "set a flag, finish the section, test the flag": only a path that answers both tests of one
        flag the same way is a path of the program;
      * whether the path has passed the `; vc_loop_exit` marker the source puts on the exits of the cell loop: such a path
        does not re-enter the loop (hipcc routes it through the loop's latch block, whose own test -- i0 + NBUF > i >= ncell
        -- leaves the loop; that arithmetic fact about the source is the one thing the audit is TOLD instead of proving).

    Every (block, state) pair reachable under these rules is walked once: fall-through, taken branches, back edges and blocks
    placed out of line alike.  Everything else is over-approximated (unknown flag -> both branch directions), so the audit can
    flag a safe program but not pass one in which some instruction outside our asm statements touches a tuple in flight."""
    ins, labels_at = [], {}
    in_asm = False
    for i, l in enumerate(lines):
        if ";;#ASMSTART" in l:
            in_asm = True
            continue
        if ";;#ASMEND" in l:
            in_asm = False
            continue
        m = LABEL.match(l)
        if m:
            labels_at[m.group(1)] = len(ins)
            continue
        if in_asm and EXIT_MARK in l:
            ins.append((i, "; " + EXIT_MARK, True))
            continue
        code = l.split(";")[0]
        if not code.strip() or code.strip().startswith("."):
            continue
        ins.append((i, code, in_asm))
    leaders = {0} | set(labels_at.values())
    for k, (_, code, _) in enumerate(ins):
        if BRANCH.match(code) or code.strip().startswith("s_endpgm"):
            leaders.add(k + 1)
    starts = sorted(x for x in leaders if x < len(ins))
    block_of = {st: (st, (starts[n + 1] if n + 1 < len(starts) else len(ins))) for n, st in enumerate(starts)}

    def static_succ(st):
        a, b = block_of[st]
        out = [b] if b < len(ins) else []
        c = ins[b - 1][1].strip()
        if c.startswith("s_endpgm"):
            return []
        m = BRANCH.match(ins[b - 1][1])
        if m:
            if c.startswith("s_branch"):
                out = []
            tgt = labels_at.get(m.group(1) or m.group(2))
            if tgt is not None and tgt < len(ins):
                out.append(tgt)
        return out

    # the cell loop = the strongly connected component of blocks that holds asm loads; its header = the member entered from
    # outside
    succ = {st: static_succ(st) for st in starts}
    index, low, on, stack, sccs, counter = {}, {}, set(), [], [], [0]
    for root in starts:
        if root in index:
            continue
        work = [(root, iter(succ[root]))]
        index[root] = low[root] = counter[0]; counter[0] += 1; stack.append(root); on.add(root)
        while work:
            v, it = work[-1]
            adv = False
            for w in it:
                if w not in index:
                    index[w] = low[w] = counter[0]; counter[0] += 1; stack.append(w); on.add(w)
                    work.append((w, iter(succ[w])))
                    adv = True
                    break
                if w in on:
                    low[v] = min(low[v], index[w])
            if adv:
                continue
            work.pop()
            if work:
                low[work[-1][0]] = min(low[work[-1][0]], low[v])
            if low[v] == index[v]:
                comp = set()
                while True:
                    w = stack.pop(); on.discard(w); comp.add(w)
                    if w == v:
                        break
                sccs.append(comp)
    headers = set()
    for comp in sccs:
        if len(comp) < 2 and not any(st in succ[st] for st in comp):
            continue
        has_load = any(ins[k][2] and LOAD.match(ins[k][1]) for st in comp for k in range(*block_of[st]))
        if has_load:
            headers |= {st for st in comp if any(st in succ[p] and p not in comp for p in starts)}

    violations, wait_counts, n_loads, end_inflight, overflow = {}, set(), set(), 0, False
    seen = set()
    work = [(0, (), False, (), None)]        # block, queue, passed an exit marker, known flags ((a, b, value), ...), vcc ('z' | 'nz' | None)
    while work:
        item = work.pop()
        if item in seen:
            continue
        seen.add(item)
        st, state, exiting, flags_t, vcc = item
        flags = {(a_, b_): v_ for a_, b_, v_ in flags_t}
        a, b = block_of[st]
        q = list(state)
        nxt = [b] if b < len(ins) else []
        refined = []             # successors that carry a pinned boolean: (block, flags)
        for k in range(a, b):
            i, code, asm = ins[k]
            c = code.strip()
            if asm:
                if EXIT_MARK in c:
                    exiting = True
                    continue
                m = LOAD.match(code)
                if m:
                    q.append(frozenset(regs_of(m.group(2))))
                    n_loads.add(i)
                    continue
                m = WAIT.match(code)
                if m:
                    n = int(m.group(1))
                    if n > 0:
                        wait_counts.add(n)
                    del q[: max(0, len(q) - n)]
                continue
            if q:
                bad = regs_of(code) & set().union(*q)
                if bad:
                    violations.setdefault(i, (i, c, sorted(bad)))
            if c.startswith("s_endpgm"):
                end_inflight = max(end_inflight, len(q))
                nxt = []
                continue
            mv = VCC_BR.match(code)
            if mv:
                tgt = labels_at.get(mv.group(2))
                taken = [tgt] if tgt is not None and tgt < len(ins) else []
                want = mv.group(1)                      # 'nz' or 'z': branch taken when vcc is that
                if vcc in ("z", "nz"):
                    nxt = taken if vcc == want else nxt
                elif isinstance(vcc, tuple):            # vcc = exec & [~]boolean of unknown value: split, pin the boolean
                    op, pair = vcc
                    val_if_nz = -1 if op == "and" else 0
                    pinned = {"nz": val_if_nz, "z": -1 - val_if_nz}
                    split = [(x, pinned[want]) for x in taken] + [(x, pinned["z" if want == "nz" else "nz"]) for x in nxt]
                    nxt = []
                    for x, val in split:
                        f2 = dict(flags)
                        f2[pair] = val
                        refined.append((x, f2))
                else:
                    nxt = nxt + taken
                continue
            m = BRANCH.match(code)
            if m:
                tgt = labels_at.get(m.group(1) or m.group(2))
                if c.startswith("s_branch"):
                    nxt = []
                if tgt is not None and tgt < len(ins):
                    nxt.append(tgt)
                continue
            m = SMOV.match(code)
            if m:
                flags[(int(m.group(1)), int(m.group(2)))] = int(m.group(3))
                continue
            m = SCSEL.match(code)
            if m:
                flags[(int(m.group(1)), int(m.group(2)))] = "U"
                if isinstance(vcc, tuple) and vcc[1] == (int(m.group(1)), int(m.group(2))):
                    vcc = None
                continue
            m = VCC_AND.match(code)
            if m:
                pair = (int(m.group(2)), int(m.group(3)))
                val = flags.get(pair)
                if val is None:
                    vcc = None
                elif val == "U":
                    vcc = (m.group(1), pair)
                elif m.group(1) == "and":
                    vcc = "nz" if val == -1 else "z"      # exec is never 0 in straight-line uniform code
                else:
                    vcc = "nz" if val == 0 else "z"
                continue
            m = SDST.match(code)
            if m:
                tok = m.group(1)
                if tok.startswith("vcc"):
                    vcc = None
                else:
                    hit = sregs_of(tok)
                    for key in [k_ for k_ in flags if set(range(k_[0], k_[1] + 1)) & hit]:
                        del flags[key]
                        if isinstance(vcc, tuple) and vcc[1] == key:
                            vcc = None
        if len(q) > 64:          # loads issued around a cycle without a wait: the queue would grow without bound
            overflow = True
            continue
        def key_of(f):
            return tuple(sorted(((a_, b_, v_) for (a_, b_), v_ in f.items()), key=str))
        vcc_out = vcc if vcc in ("z", "nz") or vcc is None else vcc
        for nx, f in [(x, flags) for x in nxt] + refined:
            if exiting and nx in headers:
                continue          # a path that has passed an exit marker does not re-enter the cell loop
            work.append((nx, tuple(q), exiting, key_of(f), vcc_out))
    if overflow:
        violations.setdefault(-1, (-1, "a cycle issues asm loads without retiring them", []))
    return [violations[k] for k in sorted(violations)], {"asm_loads": len(n_loads), "in_flight_at_end": end_inflight,
                                                         "loop_waits": [(0, n, 0) for n in sorted(wait_counts)]}


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
