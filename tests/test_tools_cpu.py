"""The measurement tools that back DESIGN.md / profiles/r01_d stay buildable: hipcc cross-compiles them for gfx950, and
the library itself builds with each measurement-aid macro the docs name."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


@pytest.mark.parametrize("src", sorted(glob.glob(os.path.join(ROOT, "profiles", "tools", "*.hip"))))
def test_profile_tool_compiles(src, tmp_path):
    out = tmp_path / "tool.o"
    r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-c", src, "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.stat().st_size > 0


@pytest.mark.parametrize("flag", ["-DVC_STREAM_ONLY", "-DVC_DBG_TIMES", "-DVC_PF=1", "-DVC_PF_SINGLE=2", "-DVC_EPI_ROWS=1", "-DVC_RCP_MERGE=0",
                                  "-DVC_LDS_REDUCE=0", "-DVC_LB_SINGLE=3", "-DVC_NO_LOADS", "-DVC_ISSUE_PIN=0", "-DVC_NT_LOADS=0 -DVC_RCP_MERGE=1",
                                  "-DVC_WT_STORES=1 -DVC_RCP_MERGE=1", "-DVC_FOLD_LOGBETA=0", "-DVC_PW_INLINE=0 -DVC_RCP_MERGE=1",
                                  "-DVC_FOLD_LOG2E=0 -DVC_OMEGA_CS=0 -DVC_HOIST_LB=0 -DVC_NR_MERGE=0 -DVC_REC_TOUCH=0 -DVC_RCP_MERGE=1"])
def test_measurement_aid_builds(flag, tmp_path):
    """One translation unit of the likelihood kernel per macro (the full library takes too long for the CPU suite)."""
    src = os.path.join(ROOT, "velocycle_amd", "csrc", "vc_main_vfull_poisson_u16.hip" if "REDUCE" in flag or "RCP" in flag
                       else "vc_main_vu_poisson.hip")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", *flag.split(), "-c", src, "-o", str(tmp_path / "k.o")],
                       capture_output=True, text=True, cwd=os.path.dirname(src))
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.parametrize("tu", ["vc_main_vfull_nb_u16.hip", "vc_main_vu_nb_u16.hip", "vc_main_phase_nb.hip", "vc_main_vu_nb_u16_pwl.hip"])
def test_asm_count_loads_are_never_touched_in_flight(tu):
    """The likelihood kernel issues its count loads from inline asm with hand-placed waits (VC_ASM_LOADS).  hipcc does not
    model such loads, so the emitted code object is audited (profiles/tools/check_asm_loads.py): in every instantiation the
    engine can select (no scratch -- vc_finalize refuses the others) no instruction touches a destination tuple between its
    load and the wait that retires it, around the loop's back edge too; every loop wait leaves the same number of
    operations outstanding; a vmcnt(0) drain retires every tuple before the epilogue."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_asm_loads", os.path.join(ROOT, "profiles", "tools", "check_asm_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rep = mod.report_tu(tu)
    assert rep, "no vc_main_kernel instantiation found"
    selectable = [r for r in rep if r["scratch"] == 0]
    assert len(selectable) >= 15                                   # every 4-genes-per-lane kernel + the small 8-genes-per-lane ones
    for r in selectable:
        assert r["asm_loads"] > 0 and not r["problems"], r
    h1 = [r for r in selectable if "ILi1ELi0E" in r["name"] and "ELi8ELi" in r["name"]]
    assert h1, "the benchmark instantiation (H = 1, no batches, 8 genes per lane) must stay scratch-free"


def test_asm_load_audit_catches_planted_hazards():
    """The audit itself: on the benchmark instantiation it must be clean, and it must flag (a) loop waits that leave one
    operation too many outstanding, (b) missing drains, (c) a compiler-style copy of a tuple right behind its load."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("check_asm_loads", os.path.join(ROOT, "profiles", "tools", "check_asm_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines, meta = mod.kernels(mod.device_asm("vc_main_vfull_nb_u16.hip"))["_Z14vc_main_kernelILi1ELi0ELi1ELi0ELi8ELi1EEv6VcDims6VcBufs"]
    v, st = mod.audit(lines)
    assert not v and st["in_flight_at_end"] == 0 and [n for _, n, _ in st["loop_waits"]] == [4] and st["asm_loads"] >= 6
    v, _ = mod.audit([l.replace("s_waitcnt vmcnt(4)", "s_waitcnt vmcnt(5)") for l in lines])
    assert v, "a wait that leaves 5 operations outstanding went unnoticed"
    kept, in_asm = [], False
    for l in lines:
        in_asm = True if ";;#ASMSTART" in l else (False if ";;#ASMEND" in l else in_asm)
        if not (in_asm and "s_waitcnt vmcnt(0)" in l):
            kept.append(l)
    v, st = mod.audit(kept)
    assert v and st["in_flight_at_end"] > 0, "missing drains went unnoticed"
    i = [k for k, l in enumerate(lines) if mod.LOAD.match(l.split(";")[0]) and ", s[" in l][4]
    reg = re.search(r"v\[(\d+):", lines[i]).group(1)
    j = i
    while ";;#ASMEND" not in lines[j]:
        j += 1
    v, _ = mod.audit(lines[:j + 1] + [f"\tv_mov_b32_e32 v255, v{reg}"] + lines[j + 1:])
    assert len(v) == 1 and int(reg) in v[0][2]


def test_asm_load_audit_path_rules_on_synthetic_code():
    """The audit's three path rules on hand-written snippets: (1) a boolean set by s_cselect and tested twice answers both
    tests the same way (hipcc's "set a flag, finish the section, test the flag"); (2) a path through the `; vc_loop_exit`
    marker does not re-enter the loop; (3) without either, the same code IS flagged (the rules prune, they do not hide)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_asm_loads", os.path.join(ROOT, "profiles", "tools", "check_asm_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def asm(*ins):
        return ["\t;;#ASMSTART", *["\t" + i for i in ins], "\t;;#ASMEND"]

    def loop(flagged_exit, marker):
        # three rotating buffers A = v[2:3], B = v[4:5], C = v[6:7], two loads in flight across the back edge (the S+U kernel's
        # shape).  An exit between sections 1 and 2 that is followed around the latch meets section 0 with A's load in flight.
        def section(load, use):
            return asm(f"global_load_dwordx2 {load}, v1, s[2:3]") + asm("s_waitcnt vmcnt(2)") + [f"\tv_add_f32_e32 v10, {use}"]
        body = [".LBB0_1:"] + section("v[6:7]", "v2, v3") + section("v[2:3]", "v4, v5")
        body += ["\ts_cmp_lt_i32 s8, s9", "\ts_cselect_b64 s[10:11], -1, 0"]
        if flagged_exit:
            body += ["\ts_and_b64 vcc, exec, s[10:11]", "\ts_cbranch_vccnz .LBB0_2"]
            body += asm("; vc_loop_exit") if marker else []
            body += [".LBB0_2:", "\tv_mul_f32_e32 v11, v10, v10", "\ts_andn2_b64 vcc, exec, s[10:11]", "\ts_cbranch_vccnz .LBB0_3"]
        else:
            body += ["\ts_and_b64 vcc, exec, s[10:11]", "\ts_cbranch_vccz .LBB0_4"]
        body += section("v[4:5]", "v6, v7")
        body += [".LBB0_3:", "\ts_cmp_lt_i32 s12, s13", "\ts_cbranch_scc1 .LBB0_1"]
        tail = [".LBB0_4:"] + asm("s_waitcnt vmcnt(0)") + ["\tv_mov_b32_e32 v2, 0", "\ts_endpgm"]
        return asm("global_load_dwordx2 v[2:3], v1, s[2:3]") + asm("global_load_dwordx2 v[4:5], v1, s[2:3]") + body + tail

    clean = loop(flagged_exit=False, marker=False)                      # exit straight to the drain: nothing to prune
    assert not mod.audit(clean)[0]
    pruned = loop(flagged_exit=True, marker=True)                       # hipcc's form: flag + marker + latch back to the header
    assert not mod.audit(pruned)[0], mod.audit(pruned)[0]
    unmarked = loop(flagged_exit=True, marker=False)                    # the same without the marker: the latch path is walked
    assert mod.audit(unmarked)[0], "the exit path around the latch was not walked"


def test_valu_model_is_what_bench_reads():
    """profiles/valu_model.json (written by profiles/tools/valu_count.py from the code objects and the GPU run of
    valu_rate.hip) carries, for the kernels bench.py runs, the measured floor of the instruction mix per occupancy."""
    import json
    vm = json.load(open(os.path.join(ROOT, "profiles", "valu_model.json")))
    assert 1.5 < vm["mix_clock_ghz"] < 3.0
    for name in ("vc_main_kernel<1,0,vfull_nb,gpl8,u16>", "vc_main_kernel<1,0,vu_nb,gpl8,u16>", "vc_main_kernel<1,0,phase_nb,gpl8,u16>"):
        ent = vm["kernels"][name]
        floor = ent["floor_ns_per_cell_iter"]
        assert {"2", "3"} <= set(floor) and ent["genes_per_lane"] == 8
        mix = vm["mixes"][ent["mix"]]
        # the mix the GPU tool ran is the kernel's own: instruction counts within 2 %
        assert abs(mix["instr"] - ent["valu_per_cell_iter"]) <= 0.02 * ent["valu_per_cell_iter"] + 2
        # more waves never cost more per instruction, and the floor sits between 1.9 and 2.4 ns per instruction
        assert floor["3"] <= floor["2"] * 1.01
        assert 1.9 < floor["2"] / ent["valu_per_cell_iter"] < 2.4


def test_small_kernels_of_the_measured_configurations_wait_for_no_load_one_by_one(tmp_path):
    """Round 6 (profiles/r06_hist_split.md): hipcc puts `s_waitcnt vmcnt(0)` behind every load of an unrolled sequence when a select, a
    conversion or a dereference stands directly behind each load -- N dependent memory round trips instead of one (the quarter blocks'
    table prefetch: 5.8 us; K_pre's dense histogram block: 14 us of the K-particle step).  The instantiations the bench line and the
    shard sweep run must stay free of such clusters (profiles/tools/scan_serial_loads.py; one translation unit: ~ 1 min of hipcc)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("scan_serial_loads", os.path.join(ROOT, "profiles", "tools", "scan_serial_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    asm = str(tmp_path / "fused.s")
    mod.compile_asm("vc_fused_kernels.hip", asm)
    res = {mod.demangled_hint(k): v for k, v in mod.scan(asm).items()}
    all_syms = open(asm).read()
    # rows of vc_tail_spec_rows.inc (1-based): 1 vjoint_rank, 6 phase, 19 vjoint, 25 vjoint_2s -- the one-launch tail of the headline,
    # of the phase model and of configs[4], and phase B of a rank of the sharded V-joint step
    for sym, hint in (("_Z15vc_tail2_kernelILi6ELi19EE", "vc_tail2_kernel<6,19>"), ("_Z15vc_tail2_kernelILi4ELi6EE", "vc_tail2_kernel<4,6>"),
                      ("_Z15vc_tail2_kernelILi6ELi25EE", "vc_tail2_kernel<6,25>"), ("_Z17vc_phase_b_kernelILi2ELi1EE", "vc_phase_b_kernel<2,1>")):
        assert sym in all_syms, f"{hint} is not compiled any more: update this list"
        assert hint not in res, (hint, res[hint])
    # the scan itself finds a planted sequence: four loads, each waited for on its own
    planted = tmp_path / "p.s"
    planted.write_text("_Z4testv:\n" + "".join(f"\tglobal_load_dword v{i}, v[2:3], off\n\ts_waitcnt vmcnt(0)\n\tv_cvt_f64_f32_e32 v[4:5], v{i}\n"
                                               for i in range(4)) + "\ts_endpgm\n")
    assert mod.scan(str(planted)) == {"_Z4testv": [(1, 10, 4, 3)]}          # (waits BETWEEN the first and the last load)
